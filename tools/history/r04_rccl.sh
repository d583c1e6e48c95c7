#!/bin/bash
OUT=gpurun_out/r04d; mkdir -p $OUT
timeout 3000 python3 tools/soak_rccl.py 2>&1 | grep -v amdgpu.ids | tee $OUT/soak_rccl.txt | tail -20
