#!/usr/bin/env python3
"""Long run of the native exchange with R ranks on one GPU (loop-back transport) against ONE oracle table: many exchanges,
a raycast round every few of them, a garbage collection now and then -- the buffer-set rotation, the deferred frames and the
event reuse of the transport over thousands of collectives.   tools/soak_native.py [R=4] [exchanges=200] [batch=4] [flatten_variant=3]
(flatten_variant 4: the shards run the walk-free multi-camera frame)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch

import oracle as O
import voxelhashing_demo_amd as V
from voxelhashing_demo_amd import dist as vdist
from voxelhashing_demo_amd import synth

R = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4
WALK = int(sys.argv[4]) if len(sys.argv) > 4 else 3
W, H = 320, 240
kw = dict(numBuckets=1 << 14, numVoxelBlocks=1 << 13)
kinv = np.linalg.inv(synth.K_matrix(W, H).astype(np.float64)).astype(np.float32)
prims = synth.room_primitives()
nf = 60
poses = [synth.camera_loop(nf, phase=vdist.camera_phase(r, R)) for r in range(R)]
d16 = [[np.round(synth.render_room_verts(p, W, H, prims).numpy()[..., 2] * 5000.0).clip(0, 65535).astype(np.uint16) for p in poses[r]] for r in range(R)]
pre = [[O.preprocess(d, kinv)[0] for d in d16[r]] for r in range(R)]
dd = [[torch.from_numpy(d).cuda() for d in d16[r]] for r in range(R)]
torch.cuda.synchronize()
full = O.OracleTable(O.default_params(**kw), W, H, 1)
g = vdist.NativeGroup(V.default_params(**kw), W, H, 1, R, B, sensor_k_inv=kinv, key_capacity=W * H // 8 * B)
for t_ in g.tables:
    t_.set_option("flatten_variant", WALK)
g.self_check()
plan = vdist.ShardPlan(kw["numBuckets"], R)
outs = [torch.empty((H, W), dtype=torch.float32, device="cuda") for _ in range(R)]
t0 = time.time()
bad = 0
for i in range(N):
    ks = [(i * B + b) % nf for b in range(B)]
    g.step([[poses[r][k] for k in ks] for r in range(R)], [[dd[r][k] for k in ks] for r in range(R)])
    for k in ks:
        vdist.reference_multi_camera_frame(full, [poses[r][k] for r in range(R)], [pre[r][k] for r in range(R)])
    if i % 7 == 6:
        vp = [poses[r][ks[-1]] for r in range(R)]
        g.raycast(vp, outs, 4096)
        g.flush()
        torch.cuda.synchronize()
        for r in range(R):
            if not np.array_equal(outs[r].cpu().numpy().view(np.uint32), full.raycast(vp[r]).view(np.uint32)):
                bad += 1
                print(f"exchange {i}: raycast of rank {r} differs")
    if i % 50 == 49:
        g.flush()
        otab = full.hash_table()
        for r, t in enumerate(g.tables):
            lo, hi = plan.bucket_range(r)
            gtab = t.hash_table()
            if not (np.array_equal(gtab["pos"], otab["pos"][lo * 5:hi * 5]) and np.array_equal(gtab["ptr"] != -1, otab["ptr"][lo * 5:hi * 5] != -1)):
                bad += 1
                print(f"exchange {i}: shard {r} differs")
        print(f"exchange {i + 1}: {int((otab['ptr'] != -1).sum())} blocks, {time.time() - t0:.0f} s, mismatches so far {bad}", flush=True)
g.flush()
from test_sharding_cpu import check_shard_against_full
tot = sum(check_shard_against_full(t, full, *plan.bucket_range(r), 5) for r, t in enumerate(g.tables))
ov = sum(t.counters()["bin_overflow"] for t in g.tables)
assert all(t.counters()["spin_timeouts"] == 0 for t in g.tables), "a serialised launch gave up waiting"
print(f"done: {N} exchanges x {B} frames x {R} cameras, {tot} blocks bit-equal, bin overflows {ov}, mismatches {bad}")
g.close()
sys.exit(1 if bad else 0)
