#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel trace + stats of the bench command for one
# workload, then separate PMC passes (FETCH_SIZE, WRITE_SIZE cannot share a pass on gfx950),
# condensed into gpurun_out/$1/.  Copy what should be judged into profiles/ afterwards
# (tools/make_pmc_latest.py builds profiles/pmc_latest.json from the PMC summaries).
#   tools/profile_round.sh r03 [C2|C3|C2band]
# The traced command is `python3 bench.py --legs none --workload WL`: the timed windows and the
# per-dispatch profile of ONE workload, so a kernel's average is not a mix of C2 and C3 launches.
#   tools/profile_round.sh r05 C3 index "--option flatten_variant=4"     (a variant of the workload: files ..._C3index.*)
set -u
TAG=${1:-r02}; WL0=${2:-C2}; SUFFIX=${3:-}; EXTRA=${4:-}
WL=$WL0$SUFFIX
export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
T=/tmp/prof_${TAG}_${WL}; rm -rf $T
rocprofv3 --kernel-trace --stats --output-format csv -d $T/trace -- python3 bench.py --legs none --workload $WL0 $EXTRA > $OUT/bench_under_rocprof_$WL.log 2> $OUT/bench_under_rocprof_$WL.err
python3 tools/prof_summary.py stats $T/trace $OUT/kernel_stats_$WL.csv > $OUT/kernel_stats_$WL.txt
grep '^{"metric"' $OUT/bench_under_rocprof_$WL.log | tail -n 1 > $OUT/bench_under_rocprof_$WL.json       # (the compact line: stdout's last)
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $T/pmc_$C -- python3 bench.py --legs none --workload $WL0 $EXTRA --steps 100 --warmup 10 --profile-steps 0 > /dev/null 2>&1
  python3 tools/prof_summary.py pmc $T/pmc_$C $OUT/pmc_${C}_$WL.json 10 > /dev/null
done
# DRAM-side read requests beside all L2->fabric read requests: what share of the launch's reads the
# Infinity Cache answered.  TCC_EA0_RDREQ_DRAM counts requests "destined for DRAM (MC)"
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum --output-format csv -d $T/pmc_dram -- python3 bench.py --legs none --workload $WL0 $EXTRA --steps 100 --warmup 10 --profile-steps 0 > /dev/null 2>&1
python3 tools/prof_summary.py pmc $T/pmc_dram $OUT/pmc_RDREQ_DRAM_$WL.json 10 > /dev/null
if [ "$WL" = "C2" ] && [ -z "$SUFFIX" ]; then
  # raycast: VALU instructions per wave (the kernel is VALU-issue bound)
  rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $T/pmc_valu -- python3 bench.py --legs raycast --workload C2 --steps 20 --warmup 5 --profile-steps 0 > /dev/null 2>&1
  python3 tools/prof_summary.py pmc $T/pmc_valu $OUT/pmc_VALU_raycast_$WL.json 2 > /dev/null
fi
if [ "$WL" = "C2" ] && [ -z "$SUFFIX" ]; then
  # the sharded path with one rank (the code path of the N > 1 lines): bench line + PMC of its table launches
  python3 bench.py --sharded --legs none --workload C2 > $OUT/bench_sharded_world1_C2.log 2> $OUT/bench_sharded_world1_C2.err; grep '^{"metric"' $OUT/bench_sharded_world1_C2.log | tail -n 1 > $OUT/bench_sharded_world1_C2.json
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --output-format csv -d $T/pmcs_$C -- python3 bench.py --sharded --legs none --workload C2 --steps 10 --warmup 2 > /dev/null 2>&1
    python3 tools/prof_summary.py pmc $T/pmcs_$C $OUT/pmc_${C}_C2sharded.json 10 > /dev/null
  done
fi
cat $OUT/kernel_stats_$WL.txt; cat $OUT/bench_under_rocprof_$WL.json
