#!/bin/bash
OUT=gpurun_out/r04; mkdir -p $OUT
for rep in 1 2; do
for lib in voxelhashing_demo_amd/lib/alt/v_sysfirst.so voxelhashing_demo_amd/lib/alt/v_alldev.so voxelhashing_demo_amd/lib/libvoxelhash_hip.so; do
  echo -n "sharded world 1, $lib: "
  VOXELHASH_LIB=$lib python3 bench.py --sharded --legs none --workload C2 --steps 100 --warmup 10 2>/dev/null | python3 -c "import sys,json; r=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(r['value'], r['roofline']['us_per_launch'], r['host_enqueue_ms_per_step'])"
done; done 2>&1 | tee $OUT/sharded_events2.txt
