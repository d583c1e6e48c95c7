#!/bin/bash
# One GPU-box session: the gpu test suite, the default bench line, and the profile passes.
#   tools/gpu_round.sh r02 [skip-tests]
set -u
TAG=${1:-r02}
OUT=gpurun_out/$TAG; mkdir -p $OUT
if [ "${2:-}" != "skip-tests" ]; then
  timeout 1800 python -m pytest tests -m gpu -q --durations=20 -x > $OUT/gputest.log 2>&1
  echo "pytest exit $?" >> $OUT/gputest.log
  tail -40 $OUT/gputest.log
fi
timeout 900 python bench.py > $OUT/bench_default.log 2> $OUT/bench_default.err; echo "bench exit $?"
tail -n 1 $OUT/bench_default.log > $OUT/bench_default.json; cp bench_detail.json $OUT/bench_default_detail.json; tail -c 3000 $OUT/bench_default.log
timeout 900 bash tools/profile_round.sh $TAG C2 > $OUT/profile_C2.log 2>&1; tail -25 $OUT/profile_C2.log
timeout 900 bash tools/profile_round.sh $TAG C3 > $OUT/profile_C3.log 2>&1; tail -12 $OUT/profile_C3.log
timeout 900 bash tools/profile_round.sh $TAG C2band > $OUT/profile_C2band.log 2>&1; tail -6 $OUT/profile_C2band.log
timeout 600 bash tools/pmc_raycast_quick.sh $TAG 1 > $OUT/pmc_raycast.log 2>&1; tail -20 $OUT/pmc_raycast.log
timeout 300 python3 tools/raycast_stamps.py > $OUT/raycast_stamps.txt 2>&1; tail -12 $OUT/raycast_stamps.txt
timeout 300 bash tools/trace_sharded.sh $TAG > $OUT/trace_sharded_tail.txt 2>&1; tail -4 $OUT/trace_sharded_tail.txt
