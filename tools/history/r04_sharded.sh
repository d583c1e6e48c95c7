#!/bin/bash
OUT=gpurun_out/r04; mkdir -p $OUT
timeout 900 python3 -m pytest tests/test_gpu_dist_native.py "tests/test_gpu_bench_paths.py::test_c2_sharded_native_exchange" tests/test_gpu_dist_loopback.py -x -q --timeout 600 2>&1 | tail -3
for div in 1 2 4 8; do
  echo "gen CU divisor $div"
  VOXELHASH_GEN_CU_DIVISOR=$div python3 bench.py --sharded --legs none --workload C2 --steps 100 --warmup 10 2>/dev/null | python3 -c "import sys,json; r=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(r['value'], r['roofline']['us_per_launch'], r['host_enqueue_ms_per_step'])"
done 2>&1 | tee $OUT/sharded_gen_cu.txt
python3 bench.py --legs none --workload C2 2>/dev/null | python3 -c "import sys,json; r=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('unsharded', r['value'], r['roofline']['us_per_launch'])" | tee -a $OUT/sharded_gen_cu.txt
bash tools/trace_sharded.sh r04 > /dev/null 2>&1; cat gpurun_out/r04/timeline_C2sharded.txt
