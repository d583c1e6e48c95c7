#!/bin/bash
set -u
OUT=gpurun_out/r06; mkdir -p $OUT
timeout 1500 python3 -m pytest tests -m gpu -x -q --durations=40 > $OUT/gputest_durations3.log 2>&1; echo "pytest exit $?"
tail -48 $OUT/gputest_durations3.log | cut -c1-180
timeout 900 bash tools/r06_index_roles.sh C2 2>&1 | tail -10
