#!/bin/bash
# round 6: the walk-free frame (flatten_variant 4) role by role on C2 -- launch time (HIP events, bench.py --legs none) and HBM-side
# traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes) with roles of the pipelined launch switched off in a
# diagnostics build (-DVH_DEBUG_SKIP_ROLES: bit 1 TSDF update, bit 2 claim, bit 3 walk; the walk off also empties the update's list)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r06; mkdir -p $OUT
LIBV=voxelhashing_demo_amd/lib/alt/v_skip.so
WL=${1:-C2}
echo "# $WL walk-free frame, roles switched off (diagnostics build): launch us (HIP events) and counter bytes per launch" | tee $OUT/index_roles_$WL.txt
for SK in 0 2 4 8 6 10 12 14; do
  VOXELHASH_LIB=$LIBV timeout 300 python3 bench.py --legs none --workload $WL --option flatten_variant=4 --option debug_skip_roles=$SK > /tmp/b.out 2>/dev/null
  US=$(python3 -c "import json;print(json.load(open('bench_detail.json'))['roofline']['us_per_launch'])")
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$C
    VOXELHASH_LIB=$LIBV rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_$C -- python3 bench.py --legs none --workload $WL --option flatten_variant=4 --option debug_skip_roles=$SK --steps 100 --warmup 10 --profile-steps 0 > /dev/null 2>&1
    python3 tools/prof_summary.py pmc /tmp/pmc_$C /tmp/pmc_$C.json 10 > /dev/null
  done
  python3 - <<PY | tee -a $OUT/index_roles_$WL.txt
import json
f=json.load(open('/tmp/pmc_FETCH_SIZE.json')); w=json.load(open('/tmp/pmc_WRITE_SIZE.json'))
k=[x for x in f if x.startswith('frame_pipelined_kernel')]
k=max(k, key=lambda x: f[x]['FETCH_SIZE'].get('count', 0)) if k else None
fk=f[k]['FETCH_SIZE']['mean'] if k else 0; wk=w.get(k,{}).get('WRITE_SIZE',{}).get('mean',0) if k else 0
print("debug_skip_roles=$SK: launch $US us; fetch %.0f KB x2 + write %.0f KB = %.2f MB per launch" % (fk, wk, (2*fk+wk)*1024/1e6))
PY
done
