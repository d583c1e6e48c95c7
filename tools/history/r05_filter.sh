#!/bin/bash
# round 5: the claim filter (option claim_filter 0 / 1), same box: parity, then the pipelined launch of both walks on C2 / C2band / C3 / C5table
set -u
OUT=gpurun_out/r05_filter; mkdir -p $OUT; rm -f $OUT/ab_filter.txt
timeout 1200 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_parity.py tests/test_gpu_sequences.py tests/test_gpu_bench_paths.py -q -x > $OUT/tests.log 2>&1; echo "pytest exit $?" >> $OUT/tests.log; tail -5 $OUT/tests.log
for WL in C2 C2band C3 C5table; do
  for FV in 3 4; do
    [ "$WL" = "C2band" ] && [ "$FV" = "4" ] && continue
    echo "== $WL flatten_variant=$FV" | tee -a $OUT/ab_filter.txt
    timeout 600 python3 tools/ab_kernels.py --set flatten_variant=$FV --option claim_filter --values 0 1 --workload $WL --pipeline 1 --batch 8 --per-round 48 --rounds 6 2>&1 | grep -v amdgpu | tee -a $OUT/ab_filter.txt
  done
done
