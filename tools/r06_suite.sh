#!/bin/bash
# the -m gpu suite (not stopping at the first failure) and the driver's bench command in one box session
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06b
timeout 1500 python -m pytest tests -m gpu -q --durations=15 > gpurun_out/r06b/gputest.log 2>&1
echo "pytest exit $?" >> gpurun_out/r06b/gputest.log
tail -25 gpurun_out/r06b/gputest.log
