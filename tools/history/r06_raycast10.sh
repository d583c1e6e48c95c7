#!/bin/bash
set -u
OUT=gpurun_out/r06; mkdir -p $OUT
q() { python3 -c "
import json
r=json.load(open('bench_detail.json')); rc=r['raycast']; print(rc['kernel_us'], rc['variants_kernel_us'], 'occ', r['config']['occupied_blocks'])"; }
{
for O in "" "--option raycast_beam=2"; do
  echo -n "C3 full model (2000 poses), hybrid lib $O: "; timeout 600 python3 bench.py --workload C3 --legs raycast --steps 20 --warmup 5 $O > /dev/null 2>&1; q
done
for F in 300 1000; do for O in "" "--option raycast_beam=2"; do
  echo -n "C3 first $F poses $O: "; timeout 600 python3 bench.py --workload C3 --frames $F --legs raycast --steps 20 --warmup 5 $O > /dev/null 2>&1; q
done; done
echo -n "C2 bench leg: "; timeout 600 python3 bench.py --workload C2 --legs raycast --steps 20 --warmup 5 > /dev/null 2>&1; q
} | tee $OUT/raycast_c3_hybrid.txt
