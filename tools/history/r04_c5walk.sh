#!/bin/bash
OUT=gpurun_out/r04; mkdir -p $OUT
for w in C2 C2band; do
for sk in 0 7 4 8 0; do
  echo "$w debug_skip_roles=$sk (bit0 commit, bit1 integrate, bit2 claim, bit3 walk)"
  VOXELHASH_LIB=voxelhashing_demo_amd/lib/alt/v_skip.so python3 bench.py --workload $w --legs none --steps 200 --warmup 20 --option debug_skip_roles=$sk 2>/dev/null | python3 -c "import sys,json; r=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(r['roofline']['us_per_launch'])"
done; done 2>&1 | tee $OUT/c2_roles.txt
