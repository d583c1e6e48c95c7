#!/bin/bash
# round 5: the split raycast -- parity first, then A/B against the fused kernel and a sweep of the item launch's grid
set -u
OUT=gpurun_out/r05_split; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_raycast.py -q -x > $OUT/test_raycast.log 2>&1; echo "pytest exit $?" >> $OUT/test_raycast.log; tail -15 $OUT/test_raycast.log
timeout 600 python3 tools/ab_raycast.py --option raycast_split --values 0 1 > $OUT/ab_split.txt 2>&1; cat $OUT/ab_split.txt
timeout 600 python3 tools/ab_raycast.py --option raycast_items_grid --values 320 640 1280 2560 3600 5120 10240 > $OUT/ab_items_grid.txt 2>&1; cat $OUT/ab_items_grid.txt
