#!/bin/bash
# round 5: role switch-off of the REFERENCE-walk pipelined launch with the model BUILT first (tools/ab_kernels.py: the option is
# toggled after the run-in, unlike bench.py --option, which never builds a model when the claim tiles are off from the start)
#   bit0 commit, bit1 integrate, bit2 claim, bit3 walk, bit4 claim tiles end before their probes
set -u
OUT=gpurun_out/r05_c3span; mkdir -p $OUT; rm -f $OUT/roles_ref.txt
for WL in C3 C5table; do
  echo "== $WL (generic build: lean_kernels=0; vertex-map frames, 60-frame model)" | tee -a $OUT/roles_ref.txt
  VOXELHASH_LIB=voxelhashing_demo_amd/lib/alt/v_skip.so timeout 900 python3 tools/ab_kernels.py --set lean_kernels=0 --option debug_skip_roles --values 0 2 4 6 8 10 12 --workload $WL --pipeline 1 --batch 8 --per-round 48 --rounds 5 2>&1 | grep -v amdgpu | tee -a $OUT/roles_ref.txt
done
