#!/bin/bash
# round 6, first call: the driver's own bench command (compact line), then the whole -m gpu suite with per-test durations
set -u
OUT=gpurun_out/r06; mkdir -p $OUT
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.out 2> $OUT/bench_driver.err; echo "bench exit $?"
tail -c 5000 $OUT/bench_driver.out
cp bench_detail.json $OUT/bench_driver_detail.json
timeout 1500 python3 -m pytest tests -m gpu -x -q --durations=60 > $OUT/gputest_durations.log 2>&1; echo "pytest exit $?"
tail -75 $OUT/gputest_durations.log
