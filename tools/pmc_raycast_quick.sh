#!/bin/bash
# Counter passes over the raycast kernel of tools/raycast_only.py (one rocprofv3 --pmc run per group; summaries
# to gpurun_out/$1).  Quicker than tools/pmc_raycast.sh (which drives bench.py).  $2: raycast_mode (default 1).
TAG=${1:-rc}; MODE=${2:-1}
OUT=gpurun_out/$TAG; mkdir -p $OUT
T=/tmp/prof_$TAG; mkdir -p $T
export TMPDIR=/tmp
i=0
for G in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD" \
         "SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_FLAT"; do
  i=$((i+1))
  rocprofv3 --pmc $G --output-format csv -d $T/g$i -- python3 tools/raycast_only.py 20 $MODE > $OUT/g$i.log 2>&1
  python3 tools/prof_summary.py pmc $T/g$i $OUT/pmc_raycast_m${MODE}_g$i.json 2 > /dev/null 2>&1
done
python3 - <<PY
import json, glob
tot = {}
for f in sorted(glob.glob("$OUT/pmc_raycast_m${MODE}_g*.json")):
    for k, v in json.load(open(f)).items():
        if "raycast" in k:
            for c, st in v.items():
                tot.setdefault(k, {})[c] = st["mean"]
for k, v in tot.items():
    w = v.get("SQ_WAVES", 1.0)
    print(k)
    for c, m in sorted(v.items()):
        print(f"   {c:28s} {m:14.1f}   per wave {m / w:10.1f}")
json.dump(tot, open("$OUT/pmc_raycast_m${MODE}.json", "w"), indent=1, sort_keys=True)
PY
