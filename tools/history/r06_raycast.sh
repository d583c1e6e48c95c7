#!/bin/bash
# round 6: the shared cooperative raycast -- parity first, then same-box A/B against the library of the round's first commit
set -u
OUT=gpurun_out/r06; mkdir -p $OUT
timeout 900 python3 -m pytest tests/test_gpu_raycast.py tests/test_gpu_view.py -x -q 2>&1 | tail -15
{
for WL in C2 C3; do
  VOXELHASH_LIB=voxelhashing_demo_amd/lib/alt/v_before.so timeout 600 python3 tools/raycast_time.py --workload $WL --label before 2>&1 | grep -v amdgpu
  timeout 600 python3 tools/raycast_time.py --workload $WL --label shared 2>&1 | grep -v amdgpu
done
VOXELHASH_LIB=voxelhashing_demo_amd/lib/alt/v_before.so timeout 600 python3 tools/raycast_time.py --workload C3 --option raycast_beam=2 --label before-coop 2>&1 | grep -v amdgpu
timeout 600 python3 tools/raycast_time.py --workload C3 --option raycast_beam=2 --label shared-coop 2>&1 | grep -v amdgpu
timeout 600 python3 tools/raycast_time.py --workload C3 --option raycast_mode=0 --label fixed-step 2>&1 | grep -v amdgpu
timeout 600 python3 tools/raycast_time.py --workload C2 --normals --label shared 2>&1 | grep -v amdgpu
VOXELHASH_LIB=voxelhashing_demo_amd/lib/alt/v_before.so timeout 600 python3 tools/raycast_time.py --workload C2 --normals --label before 2>&1 | grep -v amdgpu
} | tee $OUT/raycast_ab.txt
