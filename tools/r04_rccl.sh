#!/bin/bash
OUT=gpurun_out/r04d; mkdir -p $OUT
timeout 1700 python -m pytest tests/test_gpu_dist_rccl.py -q -x --durations=8 -k cpp > $OUT/rccl_cpp_tests.log 2>&1
echo "pytest exit $?" >> $OUT/rccl_cpp_tests.log
tail -40 $OUT/rccl_cpp_tests.log | cut -c1-800
