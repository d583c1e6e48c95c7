#!/bin/bash
# round 5: the whole -m gpu suite on one box
set -u
OUT=gpurun_out/r05; mkdir -p $OUT
timeout 2700 python -m pytest tests -m gpu -q -x --durations=15 > $OUT/gputest.log 2>&1; echo "pytest exit $?" >> $OUT/gputest.log; tail -30 $OUT/gputest.log
