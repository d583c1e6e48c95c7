#!/bin/bash
# round 5: the surviving keys of a claim tile gathered into consecutive lanes before the frustum test + probe (VH_CLAIM_COMPACT, vh_alloc.hip)
set -u
OUT=gpurun_out/r05_compact; mkdir -p $OUT; rm -f $OUT/ab.txt
q() { python3 -c "
import json,sys; p=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('value', p['value'], 'us/launch', p['roofline'].get('us_per_launch'), 'frac', p['roofline']['frac'])"; }
timeout 1200 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_parity.py tests/test_gpu_sequences.py tests/test_gpu_configs.py -q -x 2>&1 | tail -2
for i in 1 2; do
for args in "--workload C2" "--workload C2 --option flatten_variant=4" "--workload C3" "--workload C3 --option flatten_variant=4"; do
  echo -n "$args  before: " | tee -a $OUT/ab.txt; VOXELHASH_LIB=voxelhashing_demo_amd/lib/alt/v_nocompact.so python3 bench.py --legs none --no-cpu-baseline $args 2>/dev/null | q | tee -a $OUT/ab.txt
  echo -n "$args  compact: " | tee -a $OUT/ab.txt; python3 bench.py --legs none --no-cpu-baseline $args 2>/dev/null | q | tee -a $OUT/ab.txt
done; done
