#!/usr/bin/env python3
"""One-off soak of the sharded path: 4 HIP shards on one GPU (in-process exchange), 60 multi-camera
frames in batches of 3 with band allocation, collection every 5 exchanges, a raycast over the shards
every 5 exchanges -- against ONE unsharded oracle table driven through the step-level calls."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import oracle as O
import voxelhashing_demo_amd as V
from voxelhashing_demo_amd import dist as vdist, synth
from test_sharding_cpu import check_shard_against_full

W, H, world, batch, steps = 320, 240, 4, 3, 20
kw = dict(numBuckets=1 << 14, numVoxelBlocks=1 << 14)
plan = vdist.ShardPlan(kw["numBuckets"], world)
shards = [vdist.HipShard(V.default_params(**kw), W, H, 1, plan, r, W * H, batch=batch) for r in range(world)]
views = [vdist.HipViewTable(V.default_params(**kw), W, H, 1, world, 8192) for _ in range(world)]
full = O.OracleTable(O.default_params(**kw), W, H, 1)
full.set_alloc_band(0.1)
for sh in shards:
    sh.table.set_alloc_band(0.1)
prims = synth.room_primitives()
loop = [synth.camera_loop(240, phase=vdist.camera_phase(r, world)) for r in range(world)]
t0 = time.time()
for step in range(steps):
    frames = []
    for b in range(batch):
        k = (step * batch + b) * 2 % 240
        frames.append([(loop[r][k], synth.render_room_verts(loop[r][k], W, H, prims).numpy()) for r in range(world)])
    vdist.loopback_step(shards, [[frames[b][r][0] for b in range(batch)] for r in range(world)],
                        [[torch.from_numpy(frames[b][r][1]).cuda() for b in range(batch)] for r in range(world)])
    for cams in frames:
        vdist.reference_multi_camera_frame(full, [c[0] for c in cams], [c[1] for c in cams])
    if step % 5 == 4:
        # collect on the shards what their last multi-camera frame saw; the same keys on the oracle
        doomed = []
        for sh in shards:
            sh.table.synchronize()
            comp, vol = sh.table.compact(), sh.table.sdf_blocks()
            for e in comp:
                v = vol[int(e["ptr"]):int(e["ptr"]) + 512]
                seen = v["weight"] > 0
                if not seen.any() or np.abs(v["sdf"][seen]).min() >= np.float32(0.06):
                    doomed.append(tuple(e["pos"].tolist()))
            sh.table.garbage_collect(0.06)
        freed = full.delete_blocks(doomed)
        poses = [frames[-1][r][0] for r in range(world)]
        depths = vdist.loopback_raycast(shards, views, poses, capacity=8192)
        for r in range(world):
            assert np.array_equal(depths[r].view(np.uint32), full.raycast(poses[r]).view(np.uint32)), (step, r)
        total = 0
        for r, sh in enumerate(shards):
            sh.table.synchronize()
            total += check_shard_against_full(sh.table, full, *plan.bucket_range(r), 5)
            c = sh.table.counters()
            assert c["bin_overflow"] == 0 and c["heap_exhausted"] == 0 and c["spin_timeouts"] == 0
        assert total == len(full.allocated())
        print(f"exchange {step}: {total} blocks, {freed} freed, raycasts bit-equal, {time.time() - t0:.0f} s", flush=True)
print("SHARDED SOAK OK")
