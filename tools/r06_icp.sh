#!/bin/bash
# round 6: the one-launch Align against the chain of rounds -- parity tests, then timing (tools/icp_only.py)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_icp.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r06_icp_tests.txt
{
echo "== one launch"; timeout 300 python tools/icp_only.py 200
echo "== chain"; VH_ICP_PERSISTENT=0 timeout 300 python tools/icp_only.py 200
echo "== 1280x960: one launch"; ICP_SIZE=1280x960 timeout 300 python tools/icp_only.py 100
echo "== 1280x960: chain"; ICP_SIZE=1280x960 VH_ICP_PERSISTENT=0 timeout 300 python tools/icp_only.py 100
echo "== stamps"; VH_ICP_STAMPS=1 timeout 300 python tools/icp_only.py 2 2>&1 | tail -45
} > gpurun_out/r06_icp_time.txt 2>&1
cat gpurun_out/r06_icp_tests.txt gpurun_out/r06_icp_time.txt
