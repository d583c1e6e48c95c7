#!/bin/bash
OUT=gpurun_out/r04d; mkdir -p $OUT
timeout 1700 python -m pytest tests/test_gpu_dist_rccl.py -q -x --durations=8 > $OUT/rccl_tests.log 2>&1
echo "pytest exit $?" >> $OUT/rccl_tests.log
tail -40 $OUT/rccl_tests.log | cut -c1-600
for n in 2 4 8; do
VH_BENCH_SHARE_GPU=1 timeout 600 python3 bench.py --gpus $n --steps 20 --warmup 3 --legs none 2>$OUT/rccl_bench_$n.err | grep '"metric"' > $OUT/rccl_shared_gpu_bench_$n.json
python3 -c "
import json,sys
r=json.load(open('$OUT/rccl_shared_gpu_bench_$n.json')); print($n, r['value'], r['exchange_ranks'], r['config']['key_bin_overflows'], r['config']['occupied_blocks_all_ranks'])" || tail -5 $OUT/rccl_bench_$n.err
done
