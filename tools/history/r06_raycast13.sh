#!/bin/bash
set -u
OUT=gpurun_out/r06; mkdir -p $OUT
{
timeout 600 python3 tools/raycast_time.py --workload C2 --label final 2>&1 | grep -v amdgpu
for V in t3 t0 o642 o1075; do VOXELHASH_LIB=voxelhashing_demo_amd/lib/alt/v_$V.so timeout 600 python3 tools/raycast_time.py --workload C2 --label $V 2>&1 | grep -v amdgpu; done
timeout 600 python3 tools/raycast_time.py --workload C2 --label final 2>&1 | grep -v amdgpu
} | tee $OUT/raycast_ab14.txt
