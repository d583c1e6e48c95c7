#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_icp.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r06_icp_tests.txt
timeout 600 python bench.py --legs loop,next --steps 20 --warmup 5 > gpurun_out/r06_bench_closed_loop.json 2> gpurun_out/r06_bench_closed_loop.err
cat gpurun_out/r06_icp_tests.txt; tail -c 1500 gpurun_out/r06_bench_closed_loop.json
