#!/bin/bash
set -u
OUT=gpurun_out/r05_c3span; mkdir -p $OUT; rm -f $OUT/roles_ref2.txt
for WL in C5table C3; do
  echo "== $WL (lean build; vertex-map frames, 60-frame model)" | tee -a $OUT/roles_ref2.txt
  VOXELHASH_LIB=voxelhashing_demo_amd/lib/alt/v_skip.so timeout 900 python3 tools/ab_kernels.py --option debug_skip_roles --values 0 2 4 6 --workload $WL --pipeline 1 --batch 8 --per-round 48 --rounds 5 2>&1 | grep -v amdgpu | tee -a $OUT/roles_ref2.txt
  echo "== $WL (lean build; EMPTY table: no frame before the roles are switched)" | tee -a $OUT/roles_ref2.txt
  VOXELHASH_LIB=voxelhashing_demo_amd/lib/alt/v_skip.so timeout 900 python3 tools/ab_kernels.py --frames 1 --preset debug_skip_roles=7 --option debug_skip_roles --values 7 --workload $WL --pipeline 1 --batch 8 --per-round 48 --rounds 3 2>&1 | grep -v amdgpu | tee -a $OUT/roles_ref2.txt
done
