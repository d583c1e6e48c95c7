#!/bin/bash
OUT=gpurun_out/r04; mkdir -p $OUT
for w in C5table C3; do
for cs in 0 10 25 50 75; do
  echo "$w claim_span=$cs"
  python3 bench.py --workload $w --legs none --steps 25 --warmup 5 --option claim_span=$cs 2>/dev/null | python3 -c "import sys,json; r=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(r['roofline']['us_per_launch'])"
done; done 2>&1 | tee $OUT/bigtable_claim_span.txt
