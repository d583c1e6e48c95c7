#!/bin/bash
OUT=gpurun_out/r04; mkdir -p $OUT
timeout 900 python3 -m pytest tests/test_gpu_raycast.py -x -q --timeout 300 2>&1 | tail -3
tools/ab_variants.sh run tools/ab_raycast.py --option raycast_xcd --values 1 2>&1 | tee $OUT/raycast_ab_variants.txt
timeout 600 python3 tools/raycast_stamps.py 2>&1 | grep -v amdgpu.ids > $OUT/raycast_stamps.txt
grep -E "pose|cooperative form:|slowest waves:" $OUT/raycast_stamps.txt
