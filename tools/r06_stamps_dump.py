#!/usr/bin/env python3
"""Raw per-wave stamps of the cooperative raycast over bench.py's 50 poses (C2, 120 frames fused) -> gpurun_out/r06/stamps50.npz"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import voxelhashing_demo_amd as V
from bench import WORKLOADS
from voxelhashing_demo_amd import synth
wl = WORKLOADS["C2"]
poses = synth.camera_loop(500)[:120]
prims = synth.room_primitives()
t = V.SDFHashtable(V.default_params(numBuckets=wl["buckets"], numVoxelBlocks=wl["blocks"]), 640, 480, V.SEM_PINHOLE)
for p in poses:
    t.integrate(p, synth.render_room_verts(p, 640, 480, prims, device="cuda"))
depth = torch.empty((480, 640), dtype=torch.float32, device="cuda")
st = torch.zeros((4800, 8), dtype=torch.int64, device="cuda")
for i in range(5):
    t.raycast(poses[(7 * i) % 120], depth)
t.synchronize()
L = V.load()
out = []
for i in range(50):
    k = (7 * i) % 120
    assert L.vh_debug_set_raycast_stamps(t._h, st.data_ptr()) == 0
    t.raycast(poses[k], depth)
    t.synchronize()
    L.vh_debug_set_raycast_stamps(t._h, None)
    out.append(st.cpu().numpy().copy())
os.makedirs("gpurun_out/r06", exist_ok=True)
np.savez_compressed("gpurun_out/r06/stamps50.npz", stamps=np.stack(out))
print("saved", np.stack(out).shape)
