#!/bin/bash
set -u
OUT=gpurun_out/r05_rule; mkdir -p $OUT
timeout 2400 python -m pytest tests/test_gpu_dist_native.py tests/test_gpu_dist_loopback.py tests/test_gpu_dist_rccl.py tests/test_gpu_bench_contract.py tests/test_gpu_sharding.py tests/test_gpu_concurrency.py "tests/test_gpu_bench_paths.py::test_c2_sharded_native_exchange" -q --durations=5 > $OUT/tests.log 2>&1; echo "pytest exit $?" >> $OUT/tests.log; tail -12 $OUT/tests.log
python3 tools/r05_small_shard.py 2>&1 | grep buckets | tee $OUT/small_shard.txt
python3 bench.py --legs sharded 2>/dev/null | python3 -c "
import json,sys; p=json.loads([l for l in sys.stdin if l.startswith('{')][0]); s=p['sharded_world1']; print('bench: value', p['value'], 'sharded', s['value'], round(s['value']/p['value'],3))" | tee -a $OUT/small_shard.txt
