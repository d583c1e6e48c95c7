#!/bin/bash
set -u
OUT=gpurun_out/r05_check; mkdir -p $OUT; rm -f $OUT/lean2wave.txt
q() { python3 -c "
import json,sys; p=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('value', p['value'], 'us/launch', p['roofline'].get('us_per_launch'))"; }
for i in 1 2 3; do
for args in "--workload C5table" "--workload C3"; do
  echo -n "$args tile per wave: " | tee -a $OUT/lean2wave.txt; VOXELHASH_LIB=voxelhashing_demo_amd/lib/alt/v_lean2wave.so python3 bench.py --legs none --no-cpu-baseline $args 2>/dev/null | q | tee -a $OUT/lean2wave.txt
  echo -n "$args tile per workgroup: " | tee -a $OUT/lean2wave.txt; python3 bench.py --legs none --no-cpu-baseline $args 2>/dev/null | q | tee -a $OUT/lean2wave.txt
done; done
