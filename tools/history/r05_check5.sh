#!/bin/bash
set -u
OUT=gpurun_out/r05_check; mkdir -p $OUT; rm -f $OUT/igrid.txt
q() { python3 -c "
import json,sys; p=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('value', p['value'], 'us/launch', p['roofline'].get('us_per_launch'))"; }
for g in 256 512 1024 2048; do
for args in "--workload C3 --option flatten_variant=4" "--workload C5table --option flatten_variant=4"; do
  echo -n "$args pipe_integrate_grid=$g (x2 for large walk-free frames): " | tee -a $OUT/igrid.txt; python3 bench.py --legs none --no-cpu-baseline $args --option pipe_integrate_grid=$g 2>/dev/null | q | tee -a $OUT/igrid.txt
done; done
