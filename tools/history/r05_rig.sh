#!/bin/bash
# round 5: the N-rank bench line on the shared-GPU rig (one process per rank, RCCL between them, ONE GPU under all of them): the
# generation fused into the frame launches (default) against the separate launches.  Functional evidence, not a scaling figure.
set -u
OUT=gpurun_out/r05_rig; mkdir -p $OUT; rm -f $OUT/rig.txt
q() { python3 -c "
import json,sys
ls=[l for l in sys.stdin if l.startswith('{')]
if not ls: print('no line'); sys.exit()
p=json.loads(ls[0]); print('value', p['value'], 'ranks', p['exchange_ranks'].get('ranks'), 'launch us', p['roofline']['us_per_launch'], 'phases', {k: p['exchange_phases_us'][k] for k in ('generate','collectives','apply')} if p.get('exchange_phases_us') else None)"; }
for n in 2 4 8; do
  for f in 1 0; do
    echo -n "--gpus $n fused_generation=$f: " | tee -a $OUT/rig.txt
    VH_BENCH_SHARE_GPU=1 timeout 600 python3 bench.py --gpus $n --option fused_generation=$f 2>/dev/null | q | tee -a $OUT/rig.txt
  done
done
