#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_icp.py -x -q -m gpu -rs 2>&1 | tail -8 > gpurun_out/r06_icp_tests.txt
timeout 600 python -m pytest tests -q -m gpu -rs -k "skip or not skip" --co -q 2>/dev/null | tail -3 >> gpurun_out/r06_icp_tests.txt
cat gpurun_out/r06_icp_tests.txt
