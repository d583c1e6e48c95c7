#!/bin/bash
# round 5: the bench line and every profile pass the line's roofline / traffic fields are read from, one box
set -u
TAG=r05; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python bench.py > $OUT/bench_default.log 2>&1; echo "bench exit $?"
grep '"metric"' $OUT/bench_default.log > $OUT/bench_default.json
for WL in C2 C3 C2band C5table; do timeout 900 bash tools/profile_round.sh $TAG $WL > $OUT/profile_$WL.log 2>&1; tail -4 $OUT/profile_$WL.log | cut -c1-400; done
for WL in C2 C3 C5table; do timeout 900 bash tools/profile_round.sh $TAG $WL index "--option flatten_variant=4" > $OUT/profile_${WL}index.log 2>&1; tail -3 $OUT/profile_${WL}index.log | cut -c1-400; done
# the TSDF-update launch of the two-launch frame on C3: is it issue-bound?
export TMPDIR=/tmp
for G in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY"; do
  N=$(echo $G | tr ' ' '_' | cut -c1-40)
  rm -rf /tmp/prof_ci_$N
  rocprofv3 --pmc $G --output-format csv -d /tmp/prof_ci_$N -- python3 bench.py --workload C3 --legs two_launch --steps 60 --warmup 10 --profile-steps 0 > /dev/null 2>&1
  python3 tools/prof_summary.py pmc /tmp/prof_ci_$N $OUT/pmc_commit_integrate_C3_$N.json 10 > /dev/null 2>&1
done
timeout 600 bash tools/pmc_raycast_quick.sh $TAG 1 > $OUT/pmc_raycast.log 2>&1; tail -12 $OUT/pmc_raycast.log
timeout 300 python3 tools/raycast_stamps.py > $OUT/raycast_stamps.txt 2>&1; grep "^pose" $OUT/raycast_stamps.txt
timeout 300 bash tools/trace_sharded.sh $TAG > $OUT/trace_sharded_tail.txt 2>&1; tail -3 $OUT/trace_sharded_tail.txt
ls $OUT | wc -l
