#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
for nap in 0 1 2 3 4; do
echo "== nap $nap"; VH_ICP_NAP=$nap timeout 300 python tools/icp_only.py 500 2>&1 | grep "us per"
done
for nap in 0 1 2 3 4; do
echo "== nap $nap"; VH_ICP_NAP=$nap timeout 300 python tools/icp_only.py 500 2>&1 | grep "us per"
done
} > gpurun_out/r06_icp_nap.txt 2>&1
cat gpurun_out/r06_icp_nap.txt
