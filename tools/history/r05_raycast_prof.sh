#!/bin/bash
# round 5: per-kernel times of the split raycast (rocprofv3 kernel trace of tools/raycast_only.py)
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_split; mkdir -p $OUT
T=/tmp/prof_r05_split; rm -rf $T
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $T -- python3 $GRAFT_REPO_ROOT/tools/raycast_only.py 50 1 > $OUT/raycast_only.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/prof_summary.py stats $T $OUT/kernel_stats_raycast_split.csv > $OUT/kernel_stats_raycast_split.txt; grep -i "raycast\|Name" $OUT/kernel_stats_raycast_split.txt | head
