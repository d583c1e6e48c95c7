#!/bin/bash
set -u
OUT=gpurun_out/r06; mkdir -p $OUT
{
for V in before b1 b2 p1 before b1; do VOXELHASH_LIB=voxelhashing_demo_amd/lib/alt/v_$V.so timeout 600 python3 tools/raycast_time.py --workload C2 --label $V 2>&1 | grep -v amdgpu; done
for V in before b1; do VOXELHASH_LIB=voxelhashing_demo_amd/lib/alt/v_$V.so timeout 600 python3 tools/raycast_time.py --workload C2 --normals --label $V 2>&1 | grep -v amdgpu; done
for V in before b1; do VOXELHASH_LIB=voxelhashing_demo_amd/lib/alt/v_$V.so timeout 600 python3 tools/raycast_time.py --workload C3 --option raycast_beam=2 --label $V-coop 2>&1 | grep -v amdgpu; done
VOXELHASH_LIB=voxelhashing_demo_amd/lib/alt/v_b1.so timeout 300 python3 tools/raycast_stamps.py 2>&1 | grep -v amdgpu
} | tee $OUT/raycast_ab10.txt
