#!/usr/bin/env python3
"""Fuse a few C2 frames, then run N raycasts (for rocprofv3 --pmc runs of raycast_kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import voxelhashing_demo_amd as V
from bench import WORKLOADS
from voxelhashing_demo_amd import synth
wl = WORKLOADS["C2"]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 1
poses = synth.camera_loop(500)[:120]
prims = synth.room_primitives()
t = V.SDFHashtable(V.default_params(numBuckets=wl["buckets"], numVoxelBlocks=wl["blocks"]), 640, 480, V.SEM_PINHOLE)
for p in poses:
    t.integrate(p, synth.render_room_verts(p, 640, 480, prims, device="cuda"))
depth = torch.empty((480, 640), dtype=torch.float32, device="cuda")
t.set_option("raycast_mode", mode)
for i in range(n):
    t.raycast(poses[(7 * i) % 120], depth)
t.synchronize()
print("hits", float((depth > 0).float().mean()))
