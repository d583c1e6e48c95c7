#!/bin/bash
# round 5: what each role of the walk-free pipelined launch costs (diagnostics build -DVH_DEBUG_SKIP_ROLES: roles return at once)
#   bit0 commit, bit1 integrate, bit2 claim, bit3 walk (set = skipped; the commit role always runs: it rotates the counter sets):
#   0 the launch, 8 without the walk, 4 without the claim tiles, 2 without the TSDF update, 12 commit + TSDF update alone, 14 commit alone, 10 commit + claim, 6 commit + walk
set -u
OUT=gpurun_out/r05_index; mkdir -p $OUT
for WL in C2 C3 C5table; do
  echo "== $WL flatten_variant=4" | tee -a $OUT/roles_index.txt
  VOXELHASH_LIB=voxelhashing_demo_amd/lib/alt/v_skip.so timeout 600 python3 tools/ab_kernels.py --set flatten_variant=4 --option debug_skip_roles --values 0 8 4 2 12 14 10 6 --workload $WL --pipeline 1 --batch 8 --per-round 48 --rounds 5 2>&1 | grep -v amdgpu | tee -a $OUT/roles_index.txt
done
