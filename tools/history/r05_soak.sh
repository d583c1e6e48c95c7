#!/bin/bash
# round 5 soak: the native exchange over the loop-back transport against ONE oracle table, reference walk and walk-free frame
set -u
OUT=gpurun_out/r05; mkdir -p $OUT
(for CFG in "4 150 4 3" "2 200 8 3" "8 100 2 3" "4 150 4 4" "1 300 8 3"; do echo "== soak_native $CFG (ranks exchanges batch flatten_variant)"; timeout 1500 python3 tools/soak_native.py $CFG 2>&1 | grep -v amdgpu | tail -3; done) > $OUT/soak_native.txt 2>&1
cat $OUT/soak_native.txt
(echo "== tools/soak.py"; timeout 900 python3 tools/soak.py 2>&1 | grep -v amdgpu | tail -6) > $OUT/soak_single.txt 2>&1; cat $OUT/soak_single.txt
