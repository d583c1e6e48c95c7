#!/bin/bash
set -u
OUT=gpurun_out/r05_rig; mkdir -p $OUT; rm -f $OUT/emulate.txt
for n in 2 4 8; do for f in 1 0; do
  echo "== R=$n fused_generation=$f" | tee -a $OUT/emulate.txt
  timeout 600 python3 tools/emulate_ranks.py $n 8 C2 24 $f 2>&1 | grep -v amdgpu | tail -3 | tee -a $OUT/emulate.txt
done; done
