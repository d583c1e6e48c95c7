#!/bin/bash
set -u
OUT=gpurun_out/r05_fused; mkdir -p $OUT; rm -f $OUT/ab2.txt
timeout 1500 python -m pytest tests/test_gpu_dist_native.py tests/test_gpu_dist_loopback.py tests/test_gpu_dist_rccl.py "tests/test_gpu_bench_paths.py::test_c2_sharded_native_exchange" -q -x > $OUT/tests.log 2>&1; echo "pytest exit $?" >> $OUT/tests.log; tail -3 $OUT/tests.log
q() { python3 -c "
import json,sys; p=json.loads([l for l in sys.stdin if l.startswith('{')][0]); s=p.get('sharded_world1', p)
print('value', p['value'], 'sharded', s['value'], 'ratio', round(s['value']/p['value'],3) if 'sharded_world1' in p else '-', 'launch us', s['roofline']['us_per_launch'])"; }
echo -n "unsharded + fused groups=4: " | tee -a $OUT/ab2.txt; python3 bench.py --legs sharded 2>/dev/null | q | tee -a $OUT/ab2.txt
for v in gen8 gen2; do echo -n "fused $v: " | tee -a $OUT/ab2.txt; VOXELHASH_LIB=voxelhashing_demo_amd/lib/alt/v_$v.so python3 bench.py --sharded --legs none 2>/dev/null | q | tee -a $OUT/ab2.txt; done
echo -n "fused groups=4: " | tee -a $OUT/ab2.txt; python3 bench.py --sharded --legs none 2>/dev/null | q | tee -a $OUT/ab2.txt
echo -n "separate: " | tee -a $OUT/ab2.txt; python3 bench.py --sharded --legs none --option fused_generation=0 2>/dev/null | q | tee -a $OUT/ab2.txt
