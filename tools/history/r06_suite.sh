#!/bin/bash
# round 6: the whole -m gpu suite with durations, then the driver's bench command, then the raycast A/B against the round's first library
set -u
OUT=gpurun_out/r06; mkdir -p $OUT
timeout 1500 python3 -m pytest tests -m gpu -x -q --durations=40 > $OUT/gputest_durations2.log 2>&1; echo "pytest exit $?"
tail -48 $OUT/gputest_durations2.log | cut -c1-180
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver2.out 2> $OUT/bench_driver2.err; echo "bench exit $?"
tail -c 3500 $OUT/bench_driver2.out
cp bench_detail.json $OUT/bench_driver2_detail.json
for V in before; do VOXELHASH_LIB=voxelhashing_demo_amd/lib/alt/v_$V.so timeout 600 python3 tools/raycast_time.py --workload C2 --label $V 2>&1 | grep -v amdgpu; done
timeout 600 python3 tools/raycast_time.py --workload C2 --label pruned 2>&1 | grep -v amdgpu
timeout 300 python3 tools/raycast_stamps.py 2>&1 | grep -v amdgpu > $OUT/raycast_stamps.txt; grep "^pose" $OUT/raycast_stamps.txt
