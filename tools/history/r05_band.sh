#!/bin/bash
# round 5: the DDA band's set-up divisions as div_fixed with shared divisors (exact): parity, then before / after on one box
set -u
OUT=gpurun_out/r05_band; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_bench_paths.py tests/test_gpu_parity.py tests/test_gpu_overflow.py tests/test_gpu_dist_loopback.py -q -x -k "band or dda or Band or fresh_table" > $OUT/tests.log 2>&1; echo "pytest exit $?" >> $OUT/tests.log; tail -4 $OUT/tests.log
for WL in C2band C2bandSamples C2; do
  echo "== $WL" | tee -a $OUT/ab_band.txt
  bash tools/ab_variants.sh run tools/ab_kernels.py --option pipeline --values 1 --workload $WL --pipeline 1 --batch 8 --per-round 48 --rounds 6 2>&1 | grep "==\|pipeline=" | tee -a $OUT/ab_band.txt
done
