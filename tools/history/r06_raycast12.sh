#!/bin/bash
set -u
OUT=gpurun_out/r06; mkdir -p $OUT
timeout 900 python3 -m pytest tests/test_gpu_raycast.py tests/test_gpu_overflow.py -x -q 2>&1 | grep -E "passed|failed" | tail -2
{
for V in before final0; do VOXELHASH_LIB=voxelhashing_demo_amd/lib/alt/v_$V.so timeout 600 python3 tools/raycast_time.py --workload C2 --label $V 2>&1 | grep -v amdgpu; done
timeout 600 python3 tools/raycast_time.py --workload C2 --label resolve64 2>&1 | grep -v amdgpu
VOXELHASH_LIB=voxelhashing_demo_amd/lib/alt/v_final0.so timeout 600 python3 tools/raycast_time.py --workload C2 --label final0 2>&1 | grep -v amdgpu
timeout 600 python3 tools/raycast_time.py --workload C2 --label resolve64 2>&1 | grep -v amdgpu
VOXELHASH_LIB=voxelhashing_demo_amd/lib/alt/v_final0.so timeout 600 python3 tools/raycast_time.py --workload C3 --label final0 2>&1 | grep -v amdgpu
timeout 600 python3 tools/raycast_time.py --workload C3 --label resolve64 2>&1 | grep -v amdgpu
} | tee $OUT/raycast_ab13.txt
