#!/bin/bash
# round 5: where the index walk's workgroups sit in the walk-free launch and at which issue priority (generic build, same box)
set -u
OUT=gpurun_out/r05_index; mkdir -p $OUT; rm -f $OUT/ab_order.txt
for WL in C2 C3 C5table; do
  for FIRST in 0 1; do
    echo "== $WL flatten_variant=4 lean_kernels=0 index_walk_first=$FIRST" | tee -a $OUT/ab_order.txt
    timeout 600 python3 tools/ab_kernels.py --set flatten_variant=4 lean_kernels=0 index_walk_first=$FIRST --option index_walk_prio --values 0 1 --workload $WL --pipeline 1 --batch 8 --per-round 48 --rounds 6 2>&1 | grep -v amdgpu | tee -a $OUT/ab_order.txt
  done
done
