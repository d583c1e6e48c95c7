#!/bin/bash
set -u
OUT=gpurun_out/r05_c3span; mkdir -p $OUT; rm -f $OUT/span2.txt
q() { python3 -c "import sys,json; r=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(r['value'], r['roofline']['us_per_launch'], r['roofline']['frac'])"; }
for WL in C5table C4table C3; do
  for SP in 0 25 35 50 65 80; do
    echo -n "$WL claim_span=$SP: " | tee -a $OUT/span2.txt
    python3 bench.py --workload $WL --legs none --steps 100 --warmup 20 --option claim_span=$SP 2>/dev/null | q | tee -a $OUT/span2.txt
  done
done
