#!/bin/bash
set -u
OUT=gpurun_out/r06; mkdir -p $OUT
timeout 900 python3 -m pytest tests/test_gpu_raycast.py -x -q 2>&1 | tail -3
{
for V in before shared1; do VOXELHASH_LIB=voxelhashing_demo_amd/lib/alt/v_$V.so timeout 600 python3 tools/raycast_time.py --workload C2 --label $V 2>&1 | grep -v amdgpu; done
timeout 600 python3 tools/raycast_time.py --workload C2 --label shared-2d 2>&1 | grep -v amdgpu
timeout 300 python3 tools/raycast_stamps.py 2>&1 | grep -v amdgpu
} | tee $OUT/raycast_ab2.txt
