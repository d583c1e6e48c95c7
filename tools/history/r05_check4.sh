#!/bin/bash
set -u
OUT=gpurun_out/r05_check; mkdir -p $OUT; rm -f $OUT/slotfirst.txt
q() { python3 -c "
import json,sys; p=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('value', p['value'], 'us/launch', p['roofline'].get('us_per_launch'))"; }
python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_bench_paths.py -q -x 2>&1 | grep -E "passed|failed" | tail -2
for i in 1 2 3; do
for args in "--workload C2 --option flatten_variant=4" "--workload C2"; do
  echo -n "$args before: " | tee -a $OUT/slotfirst.txt; VOXELHASH_LIB=voxelhashing_demo_amd/lib/alt/v_before.so python3 bench.py --legs none --no-cpu-baseline $args 2>/dev/null | q | tee -a $OUT/slotfirst.txt
  echo -n "$args slot first (walk-free): " | tee -a $OUT/slotfirst.txt; python3 bench.py --legs none --no-cpu-baseline $args 2>/dev/null | q | tee -a $OUT/slotfirst.txt
done; done
