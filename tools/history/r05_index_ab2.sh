#!/bin/bash
# round 5: index-walk tile shapes (bitmap words per lane, list reservation per wave / per workgroup) x position in the grid, same box
set -u
OUT=gpurun_out/r05_index; mkdir -p $OUT; rm -f $OUT/ab_shapes.txt
for WL in C2 C3 C5table; do
  echo "== $WL flatten_variant=4 lean_kernels=0; option = index_walk_first" | tee -a $OUT/ab_shapes.txt
  bash tools/ab_variants.sh run tools/ab_kernels.py --set flatten_variant=4 lean_kernels=0 --option index_walk_first --values 0 1 --workload $WL --pipeline 1 --batch 8 --per-round 48 --rounds 5 2>&1 | grep "==\|index_walk" | tee -a $OUT/ab_shapes.txt
done
