#!/usr/bin/env python3
"""What an acknowledgement channel of the key exchange could be worth to the OWNER: the multi-camera frame launch of rank 0 at
R = 1, 2, 4, 8 (one rank at a time on one GPU, as tools/scaling_inputs.py) with its bins as they are against the same launch
reading 1/32 of each frame's records (a diagnostics build, -DVH_DEBUG_SKIP_ROLES, option debug_skip_roles 16) -- in steady
state ~99 % of the received keys are found present, which is what acknowledged keys would no longer travel for.
   VOXELHASH_LIB=voxelhashing_demo_amd/lib/alt/v_roles.so python3 tools/r05_ack_bound.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import voxelhashing_demo_amd as V
from voxelhashing_demo_amd import dist as vdist
from voxelhashing_demo_amd import synth

W, H, NB, VOX, BLOCKS, B = 640, 480, 1 << 20, 0.02, 1 << 16, 8
kinv = np.linalg.inv(synth.K_matrix(W, H).astype(np.float64)).astype(np.float32)
prims = synth.room_primitives()
nf = 64
for R in (1, 2, 4, 8):
    plan = vdist.ShardPlan(NB, R)
    cap = max(2048, -(-W * H // 16))
    shards = [vdist.HipShard(V.default_params(numBuckets=NB, numVoxelBlocks=BLOCKS, voxelSize=VOX), W, H, 1, plan, r, cap, batch=B,
                             sensor_k_inv=kinv) for r in range(R)]
    poses = [synth.camera_loop(500, phase=vdist.camera_phase(r, R))[:nf] for r in range(R)]
    depth = [[(synth.render_room_verts(p, W, H, prims, device="cuda")[..., 2] * 5000).round().clamp(0, 65535).to(torch.uint16)
              for p in poses[r]] for r in range(R)]
    torch.cuda.synchronize()

    def exchange(i):
        ks = [(i * B + b) % nf for b in range(B)]
        vdist.loopback_step(shards, [[poses[r][k] for k in ks] for r in range(R)], [[None] * B for _ in range(R)],
                            [[depth[r][k] for k in ks] for r in range(R)])

    if R == 2:                            # the switch does what it says: a model built with it allocates a fraction of the blocks
        for sh in shards:
            sh.table.set_option("debug_skip_roles", 16)
        exchange(0)
        torch.cuda.synchronize()
        few = sum(sh.table.counters()["occupied"] for sh in shards)
        for sh in shards:
            sh.table.set_option("debug_skip_roles", 0)
        exchange(0)
        torch.cuda.synchronize()
        print(f"   (check: blocks after one exchange with 1/32 of the records {few}, after the same exchange with all of them "
              f"{sum(sh.table.counters()['occupied'] for sh in shards)})")
    for i in range(nf // B + 2):          # the model: every frame of the loop seen once
        exchange(i)
    torch.cuda.synchronize()
    row = {}
    for label, roles in (("bins as received", 0), ("1/32 of the records", 16), ("bins as received (again)", 0)):
        for sh in shards:
            sh.table.set_option("debug_skip_roles", roles)
        for i in range(2):
            exchange(i)
        torch.cuda.synchronize()
        shards[0].table.set_profiling(True)
        n = 6
        for i in range(n):
            exchange(2 + i)
        torch.cuda.synchronize()
        kt = shards[0].table.kernel_times(reset=True)
        shards[0].table.set_profiling(False)
        row[label] = round(1e3 * kt["frame_pipelined_ms"] / (n * B), 2)
    c = shards[0].table.counters()
    print(f"R = {R}: rank 0's frame launch, us: {row}; blocks on rank 0: {c['occupied']}", flush=True)
    for s in shards:
        s.table.close()
