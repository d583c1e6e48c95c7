#!/bin/bash
# how much of the closed loop's drift is rounding: the same 59 frames with different partitions of the ICP sums
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
for env in "X=1" "VH_ICP_PERSISTENT=0" "VH_ICP_BLOCKS=240" "VH_ICP_BLOCKS=200" "VH_ICP_BLOCKS=160" "VH_ICP_BLOCKS=128" "VH_ICP_BLOCKS=255"; do
  echo "== $env"
  env $env timeout 300 python bench.py --legs loop --steps 5 --warmup 2 > /dev/null 2>&1
  python - <<'PY'
import json
d=json.load(open("bench_detail.json"))["closed_loop"]
print({k:d[k] for k in ("value","max_drift_mm","icp_rounds_per_frame","blocks","align_us")})
PY
done
} > gpurun_out/r06_drift.txt 2>&1
cat gpurun_out/r06_drift.txt
