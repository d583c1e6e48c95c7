#!/bin/bash
# round 6: the bench line (the driver's own command and the default one) and every profile pass the line's roofline / traffic
# fields are read from, on ONE box.  Afterwards, here: tools/make_pmc_latest.py gpurun_out/r06 C2 C3 C2band C2index C3index, and
# copy what is judged into profiles/r06_*.
set -u
TAG=r06; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.out 2> $OUT/bench_driver.err; echo "driver bench exit $?"
tail -n 1 $OUT/bench_driver.out > $OUT/bench_driver.json; cp bench_detail.json $OUT/bench_driver_detail.json
timeout 900 python3 bench.py > $OUT/bench_default.out 2> $OUT/bench_default.err; echo "default bench exit $?"
tail -n 1 $OUT/bench_default.out > $OUT/bench_default.json; cp bench_detail.json $OUT/bench_default_detail.json; wc -c $OUT/bench_default.json
for WL in C2 C3 C2band; do timeout 900 bash tools/profile_round.sh $TAG $WL > $OUT/profile_$WL.log 2>&1; tail -4 $OUT/profile_$WL.log | cut -c1-300; done
for WL in C2 C3; do timeout 900 bash tools/profile_round.sh $TAG $WL index "--option flatten_variant=4" > $OUT/profile_${WL}index.log 2>&1; tail -3 $OUT/profile_${WL}index.log | cut -c1-300; done
timeout 600 bash tools/pmc_raycast_quick.sh $TAG 1 > $OUT/pmc_raycast.log 2>&1; tail -12 $OUT/pmc_raycast.log
timeout 300 python3 tools/raycast_stamps.py 2>&1 | grep -v amdgpu > $OUT/raycast_stamps.txt; grep "^pose" $OUT/raycast_stamps.txt
timeout 300 bash tools/trace_sharded.sh $TAG > $OUT/trace_sharded_tail.txt 2>&1; tail -3 $OUT/trace_sharded_tail.txt
# roctx ranges of the entry points beside the kernels they launch (tracing row): marker + kernel trace, no counters
export TMPDIR=/tmp; rm -rf /tmp/prof_roctx
VOXELHASH_ROCTX=1 rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d /tmp/prof_roctx -- python3 bench.py --legs raycast --workload C2 --steps 20 --warmup 5 --raycast-steps 10 > /dev/null 2>&1
python3 - <<'PY' > $OUT/roctx_marker_stats.txt 2>&1
import csv, glob
for f in sorted(glob.glob("/tmp/prof_roctx/**/*marker*stats*.csv", recursive=True) + glob.glob("/tmp/prof_roctx/**/*marker_api_trace.csv", recursive=True))[:2]:
    rows = list(csv.DictReader(open(f)))
    print(f.split("/")[-1], len(rows), "rows")
    for r in rows[:12]:
        print("  ", {k: r[k] for k in list(r)[:6]})
PY
head -8 $OUT/roctx_marker_stats.txt
ls $OUT | wc -l
