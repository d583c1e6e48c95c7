#!/bin/bash
set -u
OUT=gpurun_out/r05_wavetile; mkdir -p $OUT; rm -f $OUT/span.txt
q() { python3 -c "
import json,sys; p=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('value', p['value'], 'us/launch', p['roofline'].get('us_per_launch'))"; }
for span in 25 40 55 70; do
  for args in "--workload C2" "--workload C2 --option flatten_variant=4"; do
  echo -n "$args claim_span=$span wave tiles: " | tee -a $OUT/span.txt; python3 bench.py --legs none --no-cpu-baseline $args --option claim_span=$span 2>/dev/null | q | tee -a $OUT/span.txt
  done
done
echo -n "C2 before (rule): " | tee -a $OUT/span.txt; VOXELHASH_LIB=voxelhashing_demo_amd/lib/alt/v_before.so python3 bench.py --legs none --no-cpu-baseline --workload C2 2>/dev/null | q | tee -a $OUT/span.txt
