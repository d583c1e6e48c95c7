#!/usr/bin/env python3
"""Soak of the one-launch Align: N calls on two room frames, every transform compared bit for bit with the first; every 1 000th
call a kernel of another stream holds 1 792 or 2 048 workgroup slots for 5 ms (tests/test_gpu_icp.py has the short form)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import voxelhashing_demo_amd as V
from voxelhashing_demo_amd import synth, tracking
W, H = 640, 480
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
poses, prims, K = synth.camera_loop(250), synth.room_primitives(), synth.K_matrix(W, H)
kinv = np.linalg.inv(K.astype(np.float64)).astype(np.float32)
v0 = synth.render_room_verts(poses[100], W, H, prims, device="cuda")
v1 = synth.render_room_verts(poses[101], W, H, prims, device="cuda")
tp, tn = torch.empty_like(v0), torch.empty_like(v0)
tracking.depth_to_maps(v0[..., 2].contiguous(), kinv, tp, tn)
trk = tracking.CameraTracking(W, H, K, flags=3)
first = trk.Align(v1, tp, tn).copy()
gt = V.SDFHashtable(V.default_params(numBuckets=1 << 10, numVoxelBlocks=64), 64, 48, 1)
hog, L = torch.cuda.Stream(), V.load()
bad = 0
t0 = time.perf_counter()
for i in range(n):
    if i % 1000 == 500:
        L.vh_debug_occupy(gt._h, hog.cuda_stream, 2048 if (i // 1000) % 2 else 1792, 5000)
    d = trk.Align(v1, tp, tn)
    if trk.iterations != 20 or not np.array_equal(d.view(np.uint32), first.view(np.uint32)):
        bad += 1
torch.cuda.synchronize()
print(f"{n} Aligns in {time.perf_counter() - t0:.1f} s, {bad} differed from the first (or stopped early); no time-out raised")
