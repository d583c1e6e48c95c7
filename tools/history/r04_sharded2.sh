#!/bin/bash
OUT=gpurun_out/r04; mkdir -p $OUT
for o in multi_walk_entries=8 multi_walk_entries=4 multi_walk_entries=8 multi_walk_entries=4; do
  echo "sharded world 1, $o"
  python3 bench.py --sharded --legs none --workload C2 --steps 100 --warmup 10 --option $o 2>/dev/null | python3 -c "import sys,json; r=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(r['value'], r['roofline']['us_per_launch'])"
done 2>&1 | tee $OUT/sharded_walk_entries.txt
