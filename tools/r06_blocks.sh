#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_render_blocks.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r06_blocks_tests.txt
python tools/ab_raycast.py --blocks --option raycast_beam --values 3 --rounds 4 --frames 120 2>&1 | tail -1 >> gpurun_out/r06_blocks_tests.txt
python tools/ab_raycast.py --blocks --option raycast_beam --values 3 --rounds 4 --frames 500 2>&1 | tail -1 >> gpurun_out/r06_blocks_tests.txt
cat gpurun_out/r06_blocks_tests.txt
