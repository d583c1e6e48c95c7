#!/usr/bin/env python3
"""Block silhouettes: how many (block, 8x8 tile) pairs the screen boxes of vh_blocks.hip produce on the bench's model, and how
many of them come from blocks with a corner behind the camera plane (whole-image box).  numpy restatement of block_bounds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import voxelhashing_demo_amd as V
from bench import WORKLOADS
from voxelhashing_demo_amd import synth
wl = WORKLOADS["C2"]; W, H = wl["width"], wl["height"]
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 500
poses = synth.camera_loop(500)[:frames]; prims = synth.room_primitives()
t = V.SDFHashtable(V.default_params(numBuckets=wl["buckets"], numVoxelBlocks=wl["blocks"], voxelSize=wl["voxel"]), W, H, V.SEM_PINHOLE)
for p in poses:
    t.integrate(p, synth.render_room_verts(p, W, H, prims, device="cuda"))
t.synchronize()
tab = t.hash_table(); pos = tab["pos"][tab["ptr"] != -1].astype(np.float64)
fx, fy, cx, cy = [float(a) for a in synth.intrinsics(W, H)]
vs = wl["voxel"]; tmin, tmax = 0.1, 5.0
print("blocks", len(pos))
for pi in (0, 70, 140, 210, 280, 350, 420):
    T = np.asarray(poses[pi % frames], np.float64).reshape(4, 4); Ti = np.linalg.inv(T)
    lo = pos * 8 * vs; hi = lo + 8 * vs
    corners = np.stack([np.where(np.array([(c >> a) & 1 for a in range(3)], bool), hi, lo) for c in range(8)], 1)   # [n, 8, 3]
    cam = corners @ Ti[:3, :3].T + Ti[:3, 3]
    z = cam[..., 2]; iz = 1.0 / np.maximum(z, 1e-6)
    u = fx * cam[..., 0] * iz + cx; v = fy * cam[..., 1] * iz + cy
    zmin, zmax = z.min(1), z.max(1)
    vis = ~((zmax < tmin - vs) | (zmin > tmax + vs))
    whole = vis & (zmin <= 0.05)
    proj = vis & ~whole
    x0 = np.clip(np.floor(u.min(1)) - 2, 0, W - 1); x1 = np.clip(np.ceil(np.minimum(u.max(1), 1e6)) + 2, 0, W - 1)
    y0 = np.clip(np.floor(v.min(1)) - 2, 0, H - 1); y1 = np.clip(np.ceil(np.minimum(v.max(1), 1e6)) + 2, 0, H - 1)
    off = (u.max(1) < -2) | (v.max(1) < -2) | (u.min(1) > W + 1) | (v.min(1) > H + 1)
    proj &= ~off
    tiles = ((x1 // 8 - x0 // 8 + 1) * (y1 // 8 - y0 // 8 + 1))[proj]
    print(f"pose {pi}: records {int(proj.sum() + whole.sum())}, whole-image boxes {int(whole.sum())} -> {int(whole.sum()) * (W // 8) * (H // 8)} pairs; "
          f"projected {int(proj.sum())} -> {int(tiles.sum())} pairs; per tile {(tiles.sum() + whole.sum() * (W // 8) * (H // 8)) / ((W // 8) * (H // 8)):.1f}")
