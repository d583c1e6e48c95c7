#!/bin/bash
OUT=gpurun_out/r04e; mkdir -p $OUT
( timeout 900 python3 tools/soak.py 300 2>&1 | grep -v amdgpu.ids | tail -6
  timeout 900 python3 tools/soak_native.py 2>&1 | grep -v amdgpu.ids | tail -6
  timeout 900 python3 tools/soak_rccl.py 2>&1 | grep -v amdgpu.ids | tail -6 ) | tee $OUT/soak_final.txt
