#!/bin/bash
OUT=gpurun_out/r04; mkdir -p $OUT
timeout 900 python3 -m pytest tests/test_gpu_raycast.py tests/test_gpu_view.py -x -q --timeout 300 2>&1 | tail -3 | tee $OUT/raycast3_tests.txt
tools/ab_variants.sh run tools/ab_raycast.py --option raycast_xcd --values 1 2>&1 | tee $OUT/raycast3_ab.txt
tools/ab_variants.sh run tools/ab_raycast.py --option raycast_xcd --values 1 2>&1 | tee -a $OUT/raycast3_ab.txt
