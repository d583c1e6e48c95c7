#!/bin/bash
set -u
OUT=gpurun_out/r05_check; mkdir -p $OUT; rm -f $OUT/tsdfdiv.txt
q() { python3 -c "
import json,sys; p=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('value', p['value'], 'us/launch', p['roofline'].get('us_per_launch'))"; }
python -m pytest tests/test_gpu_parity.py tests/test_gpu_sequences.py tests/test_gpu_pipeline.py tests/test_gpu_configs.py tests/test_gpu_full_size.py -q -x 2>&1 | grep -E "passed|failed" | tail -2
for i in 1 2; do
for args in "--workload C3 --option flatten_variant=4" "--workload C5table --option flatten_variant=4" "--workload C2 --option flatten_variant=4" "--workload C2" "--workload C3"; do
  echo -n "$args plain divisions: " | tee -a $OUT/tsdfdiv.txt; VOXELHASH_LIB=voxelhashing_demo_amd/lib/alt/v_before.so python3 bench.py --legs none --no-cpu-baseline $args 2>/dev/null | q | tee -a $OUT/tsdfdiv.txt
  echo -n "$args shared reciprocal: " | tee -a $OUT/tsdfdiv.txt; python3 bench.py --legs none --no-cpu-baseline $args 2>/dev/null | q | tee -a $OUT/tsdfdiv.txt
done; done
