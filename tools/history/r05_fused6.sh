#!/bin/bash
set -u
OUT=gpurun_out/r05_fused; mkdir -p $OUT; rm -f $OUT/ab6.txt
q() { python3 -c "
import json,sys; p=json.loads([l for l in sys.stdin if l.startswith('{')][0]); s=p.get('sharded_world1', p)
print('value', p['value'], 'sharded', s['value'], 'ratio', round(s['value']/p['value'],3) if 'sharded_world1' in p else '-', 'launch us', s['roofline']['us_per_launch'])"; }
timeout 900 python -m pytest tests/test_gpu_dist_native.py -q -k "fused or one_rank" 2>&1 | tail -2
for i in 1 2 3; do
for v in before prio3; do echo -n "$v: " | tee -a $OUT/ab6.txt; VOXELHASH_LIB=voxelhashing_demo_amd/lib/alt/v_$v.so python3 bench.py --sharded --legs none 2>/dev/null | q | tee -a $OUT/ab6.txt; done
echo -n "owner without a division: " | tee -a $OUT/ab6.txt; python3 bench.py --sharded --legs none 2>/dev/null | q | tee -a $OUT/ab6.txt
done
