#!/usr/bin/env python3
"""The fused key generation on SMALL tables: one rank through vh_dist_step_batch with 2^20 / 2^18 / 2^17 buckets (a rank's shard of
C2's table at R = 1 / 4 / 8 -- with one camera instead of R, so the launch is lighter than a real rank's), generation as a role of the
frame launches against launches of its own: does the role's chain outlast a launch that has little walk to hide it under?
   [VOXELHASH_LIB=...] python3 tools/r05_small_shard.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import voxelhashing_demo_amd as V
from voxelhashing_demo_amd import dist as vdist
from voxelhashing_demo_amd import synth

W, H, B, nf = 640, 480, 8, 64
kinv = np.linalg.inv(synth.K_matrix(W, H).astype(np.float64)).astype(np.float32)
prims = synth.room_primitives()
poses = synth.camera_loop(500)[:nf]
depth = [(synth.render_room_verts(p, W, H, prims, device="cuda")[..., 2] * 5000).round().clamp(0, 65535).to(torch.uint16) for p in poses]
torch.cuda.synchronize()
for nb in (1 << 21, 1 << 20, 1 << 19, 1 << 18, 1 << 17):
    row = []
    for fused in (1, 0):
        nd = vdist.NativeDist(V.default_params(numBuckets=nb, numVoxelBlocks=1 << 16, voxelSize=0.02), W, H, 1, 0, 1, B, vdist.unique_id(), sensor_k_inv=kinv)
        nd.set_option("fused_generation", fused)
        def step(i):
            ks = [(i * B + b) % nf for b in range(B)]
            nd.step([poses[k] for k in ks], [depth[k] for k in ks])
        for i in range(20):
            step(i)
        nd.flush(); torch.cuda.synchronize()
        best = 1e9
        for rep in range(5):
            t0 = time.perf_counter()
            for i in range(60):
                step(i)
            nd.flush(); torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        row.append(60 * B / best)
        nd.close()
    print(f"buckets 2^{nb.bit_length() - 1}: fused {row[0]:.0f} frames/s ({1e6 / row[0]:.2f} us per frame), separate {row[1]:.0f} ({1e6 / row[1]:.2f} us)", flush=True)
