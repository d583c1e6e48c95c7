#!/bin/bash
# where the 1.68 GB frame's time goes: the launch with roles switched off (diagnostics build -DVH_DEBUG_SKIP_ROLES)
OUT=gpurun_out/r04; mkdir -p $OUT
for w in C5table C3; do
for sk in 0 7 4 3; do
  echo "$w debug_skip_roles=$sk (bit0 commit, bit1 integrate, bit2 claim, bit3 walk)"
  VOXELHASH_LIB=voxelhashing_demo_amd/lib/alt/v_skip.so python3 bench.py --workload $w --legs none --steps 25 --warmup 5 --option debug_skip_roles=$sk 2>/dev/null | python3 -c "import sys,json; r=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(r['roofline']['us_per_launch'])"
done; done 2>&1 | tee $OUT/c5table_roles.txt
