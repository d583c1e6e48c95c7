#!/bin/bash
set -u
OUT=gpurun_out/r06; mkdir -p $OUT
timeout 900 python3 -m pytest tests/test_gpu_raycast.py -x -q 2>&1 | tail -3
{
for V in before k1o5 k2o5 k3o5 k2o6; do VOXELHASH_LIB=voxelhashing_demo_amd/lib/alt/v_$V.so timeout 600 python3 tools/raycast_time.py --workload C2 --label $V 2>&1 | grep -v amdgpu; done
timeout 600 python3 tools/raycast_time.py --workload C2 --label k1o6 2>&1 | grep -v amdgpu
for V in before k2o5 k3o5 k2o6; do VOXELHASH_LIB=voxelhashing_demo_amd/lib/alt/v_$V.so timeout 600 python3 tools/raycast_time.py --workload C3 --option raycast_beam=2 --label $V 2>&1 | grep -v amdgpu; done
VOXELHASH_LIB=voxelhashing_demo_amd/lib/alt/v_k2o5.so timeout 300 python3 tools/raycast_stamps.py 2>&1 | grep -v amdgpu | grep -A3 "pose 0"
} | tee $OUT/raycast_ab3.txt
