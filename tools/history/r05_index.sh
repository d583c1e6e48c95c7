#!/bin/bash
# round 5: the walk-free (occupancy-index) frame -- parity, then the pipelined launch on C2 / C3 / C5table against the reference's walk
set -u
OUT=gpurun_out/r05_index; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_parity.py tests/test_gpu_sequences.py -q -x > $OUT/tests.log 2>&1; echo "pytest exit $?" >> $OUT/tests.log; tail -5 $OUT/tests.log
for WL in C2 C3 C5table; do
  echo "== $WL" | tee -a $OUT/ab_index.txt
  timeout 600 python3 tools/ab_kernels.py --option flatten_variant --values 3 4 --workload $WL --pipeline 1 --batch 8 --per-round 48 --rounds 6 2>&1 | grep -v amdgpu | tee -a $OUT/ab_index.txt
done
