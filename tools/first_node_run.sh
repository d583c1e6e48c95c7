#!/bin/bash
# The FIRST run of this code on a node with N distinct MI355X (nothing here has ever crossed an xGMI link: every N > 1
# test so far ran on one GPU -- the loop-back transport, or RCCL between rank processes that call themselves hosts of their own,
# tests/test_gpu_dist_*.py).  Run from the repository root on the node; everything is written under gpurun_out/first_node/.
#
#   bash tools/first_node_run.sh            # N = 1, 2, 4, 8 on C2 (BASELINE.json configs[1]'s frames, one camera per GPU)
#   bash tools/first_node_run.sh C5         # ... on C5's frames (1920x1080, 2^24 buckets: BASELINE.json configs[4])
#
# What each step is for, what to compare, and what a failure looks like:
#
# 0. The library builds and binds RCCL:  python -c "import __graft_entry__ as g; g.build()"
#    A rank whose library cannot dlsym ncclAllToAll / ncclAllGather says so in `exchange_host` ("the native exchange was not
#    available: ...") and the line falls back to the Python host: treat that as a failed run, not as a number.
#
# 1. N = 1 twice: `python bench.py` (its leg `sharded_world1`) and `python bench.py --sharded`.  The two must agree within 5 %
#    (same code path, same frames): if they do not, the box is noisy or another job holds the GPU -- stop and find out.
#
# 2. N = 2, 4, 8: `python bench.py --gpus N` (bench.py starts `python -m torch.distributed.run` itself; the driver's own command
#    line is equivalent).  Before anything is timed every rank runs vh_dist_self_check: a known pattern through ncclAllToAll and
#    ncclAllGather, compared on the device.  Failure signatures:
#      * "vh_dist_self_check: rank r of N received k wrong words through the rccl transport"  -> the transport delivers to the
#        wrong place (a mis-sized slot, a stale IPC mapping): do not trust any number of that run;
#      * hipIpcGetMemHandle: invalid argument                     -> HSA_ENABLE_IPC_MODE_LEGACY is not 0 in the ranks' environment;
#      * a hang in init_process_group / the first all_reduce      -> the rendezvous: the launcher must use 127.0.0.1
#        (bench.py does); NCCL_SOCKET_IFNAME defaults to lo here, a site setting may override it;
#      * "N GPUs requested (--gpus N), M visible"                 -> fewer devices than ranks: RCCL refuses two ranks on one device;
#      * VH_ERR_TIMEOUT / vh_counters.spin_timeouts > 0           -> a serialised launch gave up (overflow list only; not in these runs);
#      * key_bin_overflows > 0 in the line's config               -> a bin was too small for a batch's keys (they are demanded again
#        by the next frame: the model is still right, the rate is not comparable).
#
# 3. Read the line against the prediction that was written down before any such run existed (profiles/r05_scaling_model.json,
#    tools/scaling_model.py; the line carries it as `predicted`): `value` against predicted.reference_walk.nominal.frames_per_s,
#    `exchange_phases_us` {generate, collectives, apply} against the model's {gen_us, comm_us, apply_us}.  The period of an
#    exchange is max(generate, collectives, apply) when the three streams overlap as designed; if `first_to_last` is close to
#    their SUM instead, the overlap is not happening (look at `generation_form`: "fused" means the generation rode in the frame
#    launches and is part of apply).  collectives far above the model: the per-link rate assumed (45-64 GB/s per direction) is
#    not what this node gives -- rerun tools/scaling_model.py with the measured rate.
#    `roofline.traffic` of an N > 1 line is the one-rank counter ratio applied to this rank's bytes, not a measurement.
#
# 4. `--option flatten_variant=4` repeats every N on the walk-free frame (the library's default; `value` above is the reference's
#    walk because SURVEY.md 8(d) quotes the roofline on its 20*N bytes): compare with predicted.walk_free.
set -u
WL=${1:-C2}
OUT=gpurun_out/first_node; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
python3 -c "import __graft_entry__ as g; g.build()" || exit 1
line() { tail -n 1 "$1" | python3 -c "
import sys, json
r = json.loads(sys.stdin.read())
p = (r.get('predicted') or {}).get('reference_walk', {}).get('frames_per_s')
print('   n_gpus', r['n_gpus'], 'value', r['value'], 'frames/s; predicted (nominal)', p, '; phases', r.get('exchange_phases_us'), '; form', r.get('generation_form'),
      '; ranks', (r.get('exchange_ranks') or {}).get('ranks'), 'self-check', (r.get('exchange_ranks') or {}).get('self_check'), '; bin overflows', r['config'].get('key_bin_overflows'))"; }
echo "== N = 1: bench.py (leg sharded_world1) and bench.py --sharded"
timeout 1200 python3 bench.py --workload $WL > $OUT/n1_default.out 2> $OUT/n1_default.err
tail -n 1 $OUT/n1_default.out | python3 -c "
import sys, json
r = json.loads(sys.stdin.read()); print('   value', r['value'], 'sharded_world1', r['legs'].get('sharded_world1'))"
timeout 600 python3 bench.py --sharded --legs none --workload $WL > $OUT/n1_sharded.out 2> $OUT/n1_sharded.err; line $OUT/n1_sharded.out
for N in 2 4 8; do
  for OPT in "" "--option flatten_variant=4"; do
    TAG=n${N}$(echo $OPT | tr -d ' =-' )
    echo "== N = $N $OPT"
    timeout 1200 python3 bench.py --gpus $N --legs none --workload $WL $OPT > $OUT/$TAG.out 2> $OUT/$TAG.err || { echo "   FAILED, see $OUT/$TAG.err"; tail -5 $OUT/$TAG.err; continue; }
    line $OUT/$TAG.out
    cp bench_detail.json $OUT/$TAG.detail.json
  done
done
