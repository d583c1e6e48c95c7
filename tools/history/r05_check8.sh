#!/bin/bash
set -u
OUT=gpurun_out/r05_check; mkdir -p $OUT; rm -f $OUT/waves.txt
q() { python3 -c "
import json,sys; p=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('value', p['value'], 'us/launch', p['roofline'].get('us_per_launch'))"; }
for i in 1 2; do
for args in "--workload C3 --option flatten_variant=4" "--workload C5table --option flatten_variant=4"; do
  for v in w7 w8; do echo -n "$args $v: " | tee -a $OUT/waves.txt; VOXELHASH_LIB=voxelhashing_demo_amd/lib/alt/v_$v.so python3 bench.py --legs none --no-cpu-baseline $args 2>/dev/null | q | tee -a $OUT/waves.txt; done
  echo -n "$args in-tree (6 waves per SIMD): " | tee -a $OUT/waves.txt; python3 bench.py --legs none --no-cpu-baseline $args 2>/dev/null | q | tee -a $OUT/waves.txt
done; done
