#!/bin/bash
# round 5: the claim tiles of the sensor-fed C3 / C5table launch: where they sit (claim_span), what they cost (diagnostics build)
set -u
OUT=gpurun_out/r05_c3span; mkdir -p $OUT; rm -f $OUT/span.txt
q() { python3 -c "import sys,json; r=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(r['value'], r['roofline']['us_per_launch'], r['roofline']['frac'])"; }
for WL in C3 C5table; do
  for SP in 0 50 70 85 100; do
    echo -n "$WL claim_span=$SP: " | tee -a $OUT/span.txt
    python3 bench.py --workload $WL --legs none --steps 100 --warmup 20 --option claim_span=$SP 2>/dev/null | q | tee -a $OUT/span.txt
  done
  for SK in 0 4 8; do
    echo -n "$WL debug_skip_roles=$SK (4: no claim tiles, 8: no walk): " | tee -a $OUT/span.txt
    VOXELHASH_LIB=voxelhashing_demo_amd/lib/alt/v_skip.so python3 bench.py --workload $WL --legs none --steps 100 --warmup 20 --option debug_skip_roles=$SK 2>/dev/null | q | tee -a $OUT/span.txt
  done
done
