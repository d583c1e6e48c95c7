#!/bin/bash
set -u
OUT=gpurun_out/r06; mkdir -p $OUT
timeout 900 python3 -m pytest tests/test_gpu_raycast.py -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
{
for V in before h0; do VOXELHASH_LIB=voxelhashing_demo_amd/lib/alt/v_$V.so timeout 600 python3 tools/raycast_time.py --workload C2 --label $V 2>&1 | grep -v amdgpu; done
timeout 600 python3 tools/raycast_time.py --workload C2 --label new 2>&1 | grep -v amdgpu
VOXELHASH_LIB=voxelhashing_demo_amd/lib/alt/v_h0.so timeout 600 python3 tools/raycast_time.py --workload C2 --label h0 2>&1 | grep -v amdgpu
timeout 600 python3 tools/raycast_time.py --workload C2 --label new 2>&1 | grep -v amdgpu
} | tee -a $OUT/raycast_ab12.txt
