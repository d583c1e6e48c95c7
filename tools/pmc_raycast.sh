#!/bin/bash
# Counter passes over the raycast kernel (one rocprofv3 --pmc run per group; summaries to gpurun_out/$1)
TAG=${1:-rc}
OUT=gpurun_out/$TAG; mkdir -p $OUT
T=/tmp/prof_$TAG; mkdir -p $T
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
i=0
for G in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD" \
         "SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_FLAT" \
         "TA_BUSY_avr TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
         "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $G --output-format csv -d $T/g$i -- python3 bench.py --legs raycast --workload C2 --steps 20 --warmup 5 --profile-steps 0 > $OUT/g$i.log 2>&1
  python3 tools/prof_summary.py pmc $T/g$i $OUT/pmc_g$i.json 2 > /dev/null 2>&1
  python3 - <<PY
import json
try:
    d=json.load(open("$OUT/pmc_g$i.json"))
    for k,v in d.items():
        if "raycast" in k: print(k, v)
except Exception as e:
    print("group $i failed", e)
PY
done
