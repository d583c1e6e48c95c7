#!/bin/bash
set -u
OUT=gpurun_out/r05_check; mkdir -p $OUT; rm -f $OUT/banddrain.txt
q() { python3 -c "
import json,sys; p=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('value', p['value'], 'us/launch', p['roofline'].get('us_per_launch'), 'frac', p['roofline']['frac'])"; }
python -m pytest tests/test_gpu_parity.py tests/test_gpu_sequences.py tests/test_gpu_pipeline.py -q -x -k "band or Band or dda" 2>&1 | grep -E "passed|failed" | tail -2
for i in 1 2 3; do
  echo -n "C2band test first in the drain: " | tee -a $OUT/banddrain.txt; VOXELHASH_LIB=voxelhashing_demo_amd/lib/alt/v_before.so python3 bench.py --legs none --no-cpu-baseline --workload C2band 2>/dev/null | q | tee -a $OUT/banddrain.txt
  echo -n "C2band slot first in the drain: " | tee -a $OUT/banddrain.txt; python3 bench.py --legs none --no-cpu-baseline --workload C2band 2>/dev/null | q | tee -a $OUT/banddrain.txt
done
