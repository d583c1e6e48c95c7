#!/bin/bash
# round 6: evidence for the one-launch Align -- per-round stamps, A/B against the chain of rounds, rocprofv3 kernel stats
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
{
echo "== tools/icp_only.py 300 (20-round Align, 640x480, two room frames): one launch";  timeout 300 python tools/icp_only.py 300 2>&1 | grep "us per\|translation"
echo "== VH_ICP_PERSISTENT=0: one launch per round"; VH_ICP_PERSISTENT=0 timeout 300 python tools/icp_only.py 300 2>&1 | grep "us per\|translation"
echo "== one launch";  timeout 300 python tools/icp_only.py 300 2>&1 | grep "us per\|translation"
echo "== VH_ICP_PERSISTENT=0"; VH_ICP_PERSISTENT=0 timeout 300 python tools/icp_only.py 300 2>&1 | grep "us per\|translation"
echo "== VH_ICP_STAMPS=1 (us since workgroup 0 started the round; [1] sums in registers [2] record stored [3] all records seen [4] added [5] estimate published; last workgroup: [6] round starts [7] record stored)"
VH_ICP_STAMPS=1 timeout 300 python tools/icp_only.py 1 2>&1 | grep -A40 "icp stamps round  0" | tail -41
} > gpurun_out/r06_icp_stamps.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_icp -- python3 tools/icp_only.py 200 > gpurun_out/r06_icp_rocprof.log 2>&1
python tools/kstats.py /tmp/prof_icp > gpurun_out/r06_kernel_stats_icp.txt 2>&1 || find /tmp/prof_icp -name "*kernel_stats*" | head
tail -50 gpurun_out/r06_icp_stamps.txt; head -12 gpurun_out/r06_kernel_stats_icp.txt
