#!/bin/bash
# round 5: what the claim tiles' PROBES (not the tiles) cost the launch: diagnostics build, bit 4 = claim tiles end before probing
set -u
OUT=gpurun_out/r05_c3span; mkdir -p $OUT; rm -f $OUT/noprobe.txt
for WL in C2 C3 C5table; do
  echo "== $WL (generic build: lean_kernels=0)" | tee -a $OUT/noprobe.txt
  VOXELHASH_LIB=voxelhashing_demo_amd/lib/alt/v_skip.so timeout 600 python3 tools/ab_kernels.py --set lean_kernels=0 --option debug_skip_roles --values 0 16 4 --workload $WL --pipeline 1 --batch 8 --per-round 48 --rounds 5 2>&1 | grep -v amdgpu | tee -a $OUT/noprobe.txt
done
