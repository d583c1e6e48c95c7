#!/bin/bash
set -u
OUT=gpurun_out/r05_check; mkdir -p $OUT; rm -f $OUT/lines.txt
q() { python3 -c "
import json,sys; p=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('value', p['value'], 'us/launch', p['roofline'].get('us_per_launch'), 'frac', p['roofline']['frac'])"; }
python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_parity.py tests/test_gpu_sequences.py tests/test_gpu_full_size.py tests/test_gpu_configs.py tests/test_gpu_bench_paths.py -q -x 2>&1 | grep -E "passed|failed" | tail -2
for args in "--workload C2" "--workload C3" "--workload C5table" "--workload C3 --option flatten_variant=4" "--workload C5table --option flatten_variant=4"; do
  echo -n "$args: " | tee -a $OUT/lines.txt; python3 bench.py --legs none --no-cpu-baseline $args 2>/dev/null | q | tee -a $OUT/lines.txt
done
