#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
nproc; grep -m1 "model name" /proc/cpuinfo; cat /proc/loadavg
timeout 300 python tools/icp_only.py 300 2>&1 | grep "us per"
VH_ICP_STAMPS=1 timeout 300 python tools/icp_only.py 6 2>&1 | grep "icp host"
} > gpurun_out/r06_icp_host.txt 2>&1
cat gpurun_out/r06_icp_host.txt
