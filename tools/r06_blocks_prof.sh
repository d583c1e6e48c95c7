#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_blocks -- python3 tools/ab_raycast.py --blocks --option raycast_beam --values 3 --rounds 3 --frames 500 > gpurun_out/r06_blocks_prof.log 2>&1
python tools/kstats.py /tmp/prof_blocks blocks_ > gpurun_out/r06_kernel_stats_blocks.txt 2>&1
tail -2 gpurun_out/r06_blocks_prof.log; cat gpurun_out/r06_kernel_stats_blocks.txt
