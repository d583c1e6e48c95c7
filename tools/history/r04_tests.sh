#!/bin/bash
OUT=gpurun_out/r04d; mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -q --durations=10 -x > $OUT/gputest.log 2>&1
echo "pytest exit $?" >> $OUT/gputest.log
tail -15 $OUT/gputest.log
timeout 900 python bench.py > $OUT/bench_default.log 2>&1; echo "bench exit $?"
grep '"metric"' $OUT/bench_default.log > $OUT/bench_default.json; tail -c 1500 $OUT/bench_default.log
