#!/bin/bash
set -u
OUT=gpurun_out/r05_fused; mkdir -p $OUT; rm -f $OUT/ab9.txt
q() { python3 -c "
import json,sys; p=json.loads([l for l in sys.stdin if l.startswith('{')][0]); s=p.get('sharded_world1', p)
print('value', s['value'], 'launch us', s['roofline']['us_per_launch'])"; }
python -m pytest tests/test_gpu_dist_native.py tests/test_gpu_dist_loopback.py -q 2>&1 | grep -E "passed|failed" | tail -2
for i in 1 2; do
for v in gi1 gi3 gi4 gi6; do echo -n "walk-free sharded, $v: " | tee -a $OUT/ab9.txt; VOXELHASH_LIB=voxelhashing_demo_amd/lib/alt/v_$v.so python3 bench.py --sharded --legs none --option flatten_variant=4 2>/dev/null | q | tee -a $OUT/ab9.txt; done
echo -n "walk-free sharded, in-tree (2 groups): " | tee -a $OUT/ab9.txt; python3 bench.py --sharded --legs none --option flatten_variant=4 2>/dev/null | q | tee -a $OUT/ab9.txt
echo -n "walk-free sharded, separate generation: " | tee -a $OUT/ab9.txt; python3 bench.py --sharded --legs none --option flatten_variant=4 --option fused_generation=0 2>/dev/null | q | tee -a $OUT/ab9.txt
done
