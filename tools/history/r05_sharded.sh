#!/bin/bash
set -u
OUT=gpurun_out/r05_dist; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_sharding.py tests/test_gpu_dist_native.py tests/test_gpu_dist_loopback.py "tests/test_gpu_bench_paths.py::test_c2_sharded_native_exchange" -q -x > $OUT/tests_marks.log 2>&1; echo "pytest exit $?" >> $OUT/tests_marks.log; tail -4 $OUT/tests_marks.log
for i in 1 2; do
timeout 600 python3 bench.py --legs sharded > $OUT/bench_sharded_leg.log 2>&1; grep '"metric"' $OUT/bench_sharded_leg.log | python3 -c "
import json,sys; p=json.loads(sys.stdin.read()); s=p['sharded_world1']; print('value',p['value'],'sharded',s['value'],'ratio',round(s['value']/p['value'],3),'launch',s['roofline']['us_per_launch'],'phases',{k:s['exchange_phases_us'][k] for k in ('generate','apply')})"
done
