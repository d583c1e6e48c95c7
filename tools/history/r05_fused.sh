#!/bin/bash
# round 5: the key generation as a role of the frame launches (vh_dist option fused_generation 1, the default) against the separate
# generation on its own stream (0), same box: the sharded leg with one rank, next to the unsharded value of the same run
set -u
OUT=gpurun_out/r05_fused; mkdir -p $OUT; rm -f $OUT/ab.txt
q() { python3 -c "
import json,sys; p=json.loads([l for l in sys.stdin if l.startswith('{')][0]); s=p.get('sharded_world1', p)
print('value', p['value'], 'sharded', s['value'], 'ratio', round(s['value']/p['value'],3) if 'sharded_world1' in p else '-', 'launch us', s['roofline']['us_per_launch'], 'phases', {k: s['exchange_phases_us'][k] for k in ('generate','apply','host_enqueue')})"; }
for i in 1 2; do
  echo -n "fused (default): " | tee -a $OUT/ab.txt; python3 bench.py --legs sharded 2>/dev/null | q | tee -a $OUT/ab.txt
  echo -n "separate (fused_generation=0): " | tee -a $OUT/ab.txt; python3 bench.py --sharded --legs none --option fused_generation=0 2>/dev/null | q | tee -a $OUT/ab.txt
done
