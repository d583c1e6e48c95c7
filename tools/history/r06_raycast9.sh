#!/bin/bash
set -u
OUT=gpurun_out/r06; mkdir -p $OUT
timeout 900 python3 -m pytest tests/test_gpu_raycast.py tests/test_gpu_view.py tests/test_gpu_sequences.py -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
{
for V in before b1; do VOXELHASH_LIB=voxelhashing_demo_amd/lib/alt/v_$V.so timeout 600 python3 tools/raycast_time.py --workload C2 --label $V 2>&1 | grep -v amdgpu; done
timeout 600 python3 tools/raycast_time.py --workload C2 --label final 2>&1 | grep -v amdgpu
} | tee $OUT/raycast_ab11.txt
mkdir -p gpurun_out/r06; timeout 1500 python3 -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r06/gputest_durations5.log 2>&1; echo "pytest exit $?"; tail -12 gpurun_out/r06/gputest_durations5.log | cut -c1-170
