#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel trace + stats of the default bench
# command, then separate PMC passes (FETCH_SIZE, WRITE_SIZE cannot share a pass on gfx950),
# condensed into gpurun_out/$1/.  Copy what should be judged into profiles/ afterwards.
#   tools/profile_round.sh r01 [C2|C3]
set -u
TAG=${1:-r01}; WL=${2:-C2}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
T=/tmp/prof_${TAG}_${WL}; rm -rf $T
rocprofv3 --kernel-trace --stats --output-format csv -d $T/trace -- python3 bench.py --no-cpu-baseline --workload $WL > $OUT/bench_under_rocprof_$WL.log 2>&1
python3 tools/prof_summary.py stats $T/trace $OUT/kernel_stats_$WL.csv > $OUT/kernel_stats_$WL.txt
grep '"metric"' $OUT/bench_under_rocprof_$WL.log > $OUT/bench_under_rocprof_$WL.json
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $T/pmc_$C -- python3 bench.py --no-cpu-baseline --workload $WL --steps 100 --warmup 10 --profile-steps 0 --raycast-steps 0 > /dev/null 2>&1
  python3 tools/prof_summary.py pmc $T/pmc_$C $OUT/pmc_${C}_$WL.json 10 > /dev/null
done
python3 bench.py --workload $WL > $OUT/bench_$WL.log 2>&1; grep '"metric"' $OUT/bench_$WL.log > $OUT/bench_$WL.json
cat $OUT/kernel_stats_$WL.txt; cat $OUT/bench_$WL.json
