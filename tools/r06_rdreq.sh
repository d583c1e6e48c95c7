#!/bin/bash
# round 6: what sizes are the L2 -> fabric read requests of the walk-free launch?  FETCH_SIZE x 2 (the guide's gfx950 correction) holds
# for wide streaming reads (128-byte requests tallied at 64); a launch of narrow gathers may issue smaller ones.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r06; mkdir -p $OUT
for WL in C2 C2index; do
  EX=""; [ "$WL" = "C2index" ] && EX="--option flatten_variant=4"
  rm -rf /tmp/pmc_rd
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d /tmp/pmc_rd -- python3 bench.py --legs none --workload C2 $EX --steps 100 --warmup 10 --profile-steps 0 > /dev/null 2>&1
  python3 tools/prof_summary.py pmc /tmp/pmc_rd $OUT/pmc_RDREQ_32B_$WL.json 10 > /dev/null 2>&1
  python3 - <<PY
import json
d=json.load(open("$OUT/pmc_RDREQ_32B_$WL.json"))
for k,v in d.items():
    if k.startswith("frame_pipelined"):
        print("$WL", k[-44:], {c: round(s["mean"]) for c, s in v.items()})
PY
done
