#!/bin/bash
set -u
OUT=gpurun_out/r06; mkdir -p $OUT
{
for V in before steal1 steal2 steal3 steal4 steal6 before; do VOXELHASH_LIB=voxelhashing_demo_amd/lib/alt/v_$V.so timeout 600 python3 tools/raycast_time.py --workload C2 --label $V 2>&1 | grep -v amdgpu; done
} | tee $OUT/raycast_ab6.txt
