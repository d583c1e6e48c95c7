#!/bin/bash
set -u
OUT=gpurun_out/r05_fused; mkdir -p $OUT; rm -f $OUT/ab10.txt
q() { python3 -c "
import json,sys; p=json.loads([l for l in sys.stdin if l.startswith('{')][0]); s=p.get('sharded_world1', p)
print('value', s['value'], 'launch us', s['roofline']['us_per_launch'])"; }
for i in 1 2; do
echo -n "walk-free sharded, default (fused host path, separate launches): " | tee -a $OUT/ab10.txt; python3 bench.py --sharded --legs none --option flatten_variant=4 2>/dev/null | q | tee -a $OUT/ab10.txt
echo -n "walk-free sharded, fused_generation=0: " | tee -a $OUT/ab10.txt; python3 bench.py --sharded --legs none --option flatten_variant=4 --option fused_generation=0 2>/dev/null | q | tee -a $OUT/ab10.txt
done
python -m pytest tests/test_gpu_dist_native.py tests/test_gpu_dist_loopback.py tests/test_gpu_bench_paths.py -q 2>&1 | grep -E "passed|failed" | tail -2
