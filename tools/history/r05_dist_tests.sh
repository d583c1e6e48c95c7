#!/bin/bash
set -u
OUT=gpurun_out/r05_dist; mkdir -p $OUT
timeout 2400 python -m pytest tests/test_gpu_dist_native.py tests/test_gpu_dist_loopback.py tests/test_gpu_dist_rccl.py tests/test_gpu_concurrency.py tests/test_gpu_bench_contract.py tests/test_gpu_overflow.py tests/test_gpu_sharding.py "tests/test_gpu_bench_paths.py::test_c2_sharded_native_exchange" -q --durations=8 > $OUT/tests.log 2>&1; echo "pytest exit $?" >> $OUT/tests.log; tail -25 $OUT/tests.log
timeout 600 python3 bench.py --sharded --legs none > $OUT/bench_sharded.log 2>&1; grep '"metric"' $OUT/bench_sharded.log > $OUT/bench_sharded.json; python3 -c "
import json; p=json.loads(open('$OUT/bench_sharded.json').read()); print(p['value'], p['exchange_ranks'], p['exchange_phases_us'], p['predicted']['reference_walk']['nominal'] if p['predicted'] else None)"
