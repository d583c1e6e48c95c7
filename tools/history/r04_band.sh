#!/bin/bash
OUT=gpurun_out/r04; mkdir -p $OUT
timeout 900 python3 -m pytest tests/test_gpu_overflow.py -x -q -k "ray_dda" --timeout 600 2>&1 | tail -4
for o in band_mode=0 band_mode=2; do
  echo "C2band $o"
  python3 bench.py --workload C2band --legs none --option $o 2>/dev/null | python3 -c "import sys,json; r=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(r['value'], r['roofline']['us_per_launch'], r['roofline']['frac'], r['config']['occupied_blocks'], r['roofline']['bytes_per_launch'])"
done 2>&1 | tee $OUT/band_mode_ab.txt
