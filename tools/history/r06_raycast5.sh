#!/bin/bash
set -u
OUT=gpurun_out/r06; mkdir -p $OUT
timeout 900 python3 -m pytest tests/test_gpu_raycast.py tests/test_gpu_view.py -x -q 2>&1 | tail -3
{
VOXELHASH_LIB=voxelhashing_demo_amd/lib/alt/v_before.so timeout 600 python3 tools/raycast_time.py --workload C2 --label before 2>&1 | grep -v amdgpu
timeout 600 python3 tools/raycast_time.py --workload C2 --label hybrid 2>&1 | grep -v amdgpu
VOXELHASH_LIB=voxelhashing_demo_amd/lib/alt/v_before.so timeout 600 python3 tools/raycast_time.py --workload C2 --normals --label before 2>&1 | grep -v amdgpu
timeout 600 python3 tools/raycast_time.py --workload C2 --normals --label hybrid 2>&1 | grep -v amdgpu
VOXELHASH_LIB=voxelhashing_demo_amd/lib/alt/v_before.so timeout 600 python3 tools/raycast_time.py --workload C3 --option raycast_beam=2 --label before-coop 2>&1 | grep -v amdgpu
timeout 600 python3 tools/raycast_time.py --workload C3 --option raycast_beam=2 --label hybrid-coop 2>&1 | grep -v amdgpu
timeout 600 python3 tools/raycast_time.py --workload C3 --label hybrid-default 2>&1 | grep -v amdgpu
timeout 300 python3 tools/raycast_stamps.py 2>&1 | grep -v amdgpu
} | tee $OUT/raycast_ab5.txt
