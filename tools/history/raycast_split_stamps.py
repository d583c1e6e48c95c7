#!/usr/bin/env python3
"""Per-wave timeline of the split raycast's first two launches on C2 (vh_debug_set_raycast_stamps): the list launch's waves
(start, set built, list resolved, end) and the item launch's waves (start, queues known, items taken / entered, time in
entered blocks, end) on the 100 MHz constant clock."""
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
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 3600
t.set_option("raycast_split", 1)
t.set_option("raycast_items_grid", grid)
nA = 40 * 30 * 4
nB = grid * 4
st = torch.zeros((nA + nB, 8), dtype=torch.int64, device="cuda")
for i in range(5):
    t.raycast(poses[(7 * i) % 120], depth)
t.synchronize()
L = V.load()
for k in (0, 35, 77):
    st.zero_()
    assert L.vh_debug_set_raycast_stamps(t._h, st.data_ptr()) == 0
    t.raycast(poses[k], depth)
    t.synchronize()
    L.vh_debug_set_raycast_stamps(t._h, None)
    s = st.cpu().numpy()
    a, b = s[:nA], s[nA:]
    b = b[b[:, 0] > 0]
    t0 = a[:, 0].min()
    us = lambda x: (x - t0) / 100.0
    sa, ea = us(a[:, 0]), us(a[:, 1])
    print(f"pose {k}: list launch: waves start median {np.median(sa):.1f} max {sa.max():.1f}, end mean {ea.mean():.1f} p99 {np.percentile(ea, 99):.1f} max {ea.max():.1f} us; "
          f"life mean {(ea - sa).mean():.2f}; set-up {(a[:, 2] / 100).mean():.2f}, set built {(a[:, 4] / 100).mean():.2f}, list resolved {(a[:, 5] / 100).mean():.2f} (max {(a[:, 5] / 100).max():.2f}); "
          f"list mean {a[:, 6].mean():.2f} max {a[:, 6].max()} total {a[:, 6].sum()}")
    sb, eb = us(b[:, 0]), us(b[:, 1])
    items, entered = b[:, 2] & 0xffffffff, b[:, 2] >> 32
    print(f"   item launch: {len(b)} waves, first start {sb.min():.1f}, start median {np.median(sb):.1f} max {sb.max():.1f}, end mean {eb.mean():.1f} p99 {np.percentile(eb, 99):.1f} max {eb.max():.1f} us; "
          f"life mean {(eb - sb).mean():.2f} max {(eb - sb).max():.2f}; queues known after {(b[:, 3] / 100).mean():.2f}; items per wave mean {items.mean():.2f} max {items.max()}, entered {entered.mean():.2f}; "
          f"items total {items.sum()} (counters say {b[0, 5]}), entered total {entered.sum()}; time in entry+walk per wave {(b[:, 4] / 100).mean():.2f} us, per item {(b[:, 4].sum() / 100) / max(1, items.sum()):.2f} us")
    h, _ = np.histogram(eb, bins=10, range=(sb.min(), eb.max()))
    print("   item waves ending per tenth of the launch:", h.tolist())
