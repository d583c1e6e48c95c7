#!/bin/bash
set -u
OUT=gpurun_out/r05_index; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_bench_paths.py tests/test_gpu_full_size.py tests/test_gpu_dist_loopback.py tests/test_gpu_sharding.py -q -x --durations=8 > $OUT/tests2.log 2>&1; echo "pytest exit $?" >> $OUT/tests2.log; tail -16 $OUT/tests2.log
