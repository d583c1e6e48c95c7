#!/bin/bash
set -u
OUT=gpurun_out/r06; mkdir -p $OUT
q() { python3 -c "
import sys,json
r=json.load(open('bench_detail.json')); rc=r['raycast']; print(rc['kernel_us'], rc['variants_kernel_us'], 'occ', r['config']['occupied_blocks'])"; }
{
for V in before k1o5; do
  for O in "" "--option raycast_beam=2"; do
    echo -n "C3 full model, lib $V $O: "; VOXELHASH_LIB=voxelhashing_demo_amd/lib/alt/v_$V.so timeout 600 python3 bench.py --workload C3 --legs raycast --steps 20 --warmup 5 $O > /dev/null 2>&1; q
  done
done
} | tee $OUT/raycast_c3.txt
