#!/bin/bash
# round 5: a launch tile per WAVE in the claim role of the pipelined frame (claim_tile_wave, vh_alloc.hip) against a tile per workgroup
set -u
OUT=gpurun_out/r05_wavetile; mkdir -p $OUT; rm -f $OUT/ab.txt
q() { python3 -c "
import json,sys; p=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('value', p['value'], 'us/launch', p['roofline'].get('us_per_launch'), 'frac', p['roofline']['frac'])"; }
timeout 1500 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_parity.py tests/test_gpu_sequences.py tests/test_gpu_configs.py tests/test_gpu_full_size.py -q -x 2>&1 | tail -3
for i in 1 2; do
for args in "--workload C2" "--workload C2 --option flatten_variant=4" "--workload C3" "--workload C3 --option flatten_variant=4"; do
  echo -n "$args  before: " | tee -a $OUT/ab.txt; VOXELHASH_LIB=voxelhashing_demo_amd/lib/alt/v_before.so python3 bench.py --legs none --no-cpu-baseline $args 2>/dev/null | q | tee -a $OUT/ab.txt
  echo -n "$args  wave tiles: " | tee -a $OUT/ab.txt; python3 bench.py --legs none --no-cpu-baseline $args 2>/dev/null | q | tee -a $OUT/ab.txt
done; done
