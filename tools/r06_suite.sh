#!/bin/bash
# the -m gpu suite and the driver's bench command in one box session
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06b
timeout 1500 python -m pytest tests -m gpu -q --durations=15 -x > gpurun_out/r06b/gputest.log 2>&1
echo "pytest exit $?" >> gpurun_out/r06b/gputest.log
tail -25 gpurun_out/r06b/gputest.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06b/bench_driver.log 2> gpurun_out/r06b/bench_driver.err; echo "bench exit $?"
tail -n 1 gpurun_out/r06b/bench_driver.log > gpurun_out/r06b/bench_driver.json; cp bench_detail.json gpurun_out/r06b/bench_driver_detail.json
tail -c 3000 gpurun_out/r06b/bench_driver.json
