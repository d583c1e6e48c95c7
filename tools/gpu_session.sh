#!/bin/bash
# One GPU-box session of a round: optional pytest selection, then the commands given as arguments, each logged.
#   tools/gpu_session.sh TAG "pytest args" "cmd1" "cmd2" ...
set -u
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
SEL=$1; shift
if [ -n "$SEL" ]; then
  timeout 1800 python -m pytest $SEL -q --durations=15 > $OUT/gputest.log 2>&1
  echo "pytest exit $?" >> $OUT/gputest.log
  tail -30 $OUT/gputest.log
fi
i=0
for CMD in "$@"; do
  i=$((i+1))
  echo "=== $CMD" | tee $OUT/cmd$i.log
  timeout 900 bash -c "$CMD" >> $OUT/cmd$i.log 2>&1
  echo "exit $?" >> $OUT/cmd$i.log
  tail -25 $OUT/cmd$i.log
done
