#!/bin/bash
# (a) the ceiling probe in the walk's own shapes (records per lane and workgroup, a frame-sized kernel-argument block);
# (b) the frame's walk with 4 / 8 entries per lane on C3 and on the 1.68 GB table.   Output: gpurun_out/r04/
OUT=gpurun_out/r04; mkdir -p $OUT
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o /tmp/membw tools/micro/membw.hip || exit 1
(for MB in 419 1680; do echo "== $MB MB"; /tmp/membw $MB | grep -E "ptr|dwordx4 nt"; done) > $OUT/membw_shapes.txt 2>&1
for o in walk_entries=4 walk_entries=8; do for w in C3 C5table; do
  echo "$w $o"
  python3 bench.py --workload $w --legs none --option $o 2>/dev/null | python3 -c "import sys,json; r=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(r['value'], r['roofline']['us_per_launch'], r['roofline']['frac'])"
done; done > $OUT/walk_entries_ab.txt 2>&1
cat $OUT/membw_shapes.txt $OUT/walk_entries_ab.txt
