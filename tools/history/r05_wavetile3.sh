#!/bin/bash
set -u
OUT=gpurun_out/r05_wavetile; mkdir -p $OUT; rm -f $OUT/auto.txt
q() { python3 -c "
import json,sys; p=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('value', p['value'], 'us/launch', p['roofline'].get('us_per_launch'))"; }
timeout 1500 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_bench_paths.py tests/test_gpu_full_size.py -q -x 2>&1 | tail -3 | tee $OUT/tests_tail.txt
for i in 1 2; do
for wl in C2 C3 C5table; do
  for o in 0 2; do
  echo -n "$wl flatten_variant=4 claim_wave_tiles=$o: " | tee -a $OUT/auto.txt; python3 bench.py --legs none --no-cpu-baseline --workload $wl --option flatten_variant=4 --option claim_wave_tiles=$o 2>/dev/null | q | tee -a $OUT/auto.txt
  done
done
echo -n "C2 reference walk: " | tee -a $OUT/auto.txt; python3 bench.py --legs none --no-cpu-baseline --workload C2 2>/dev/null | q | tee -a $OUT/auto.txt
done
