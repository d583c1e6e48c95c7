#!/bin/bash
# round 5: the bucket's first slot requested before the frustum test in the claim tile (VH_CLAIM_EARLY_SLOT, vh_alloc.hip)
set -u
OUT=gpurun_out/r05_earlyslot; mkdir -p $OUT; rm -f $OUT/ab.txt
q() { python3 -c "
import json,sys; p=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('value', p['value'], 'us/launch', p['roofline'].get('us_per_launch'), 'frac', p['roofline']['frac'])"; }
python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_parity.py tests/test_gpu_sequences.py -q -x 2>&1 | grep -E "passed|failed" | tail -2
for i in 1 2 3; do
for args in "--workload C2" "--workload C2 --option flatten_variant=4" "--workload C3" "--workload C5table"; do
  echo -n "$args  before: " | tee -a $OUT/ab.txt; VOXELHASH_LIB=voxelhashing_demo_amd/lib/alt/v_before.so python3 bench.py --legs none --no-cpu-baseline $args 2>/dev/null | q | tee -a $OUT/ab.txt
  echo -n "$args  early slot: " | tee -a $OUT/ab.txt; python3 bench.py --legs none --no-cpu-baseline $args 2>/dev/null | q | tee -a $OUT/ab.txt
done; done
