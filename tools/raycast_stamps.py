#!/usr/bin/env python3
"""Per-wave timeline of one cooperative raycast launch on C2 (vh_debug_set_raycast_stamps): when each wave started and ended
on the 100 MHz constant clock, how many blocks its own patch listed, how many items it walked itself and how many it took from
its workgroup's other patches -- so that the front end, the walk per item and the balance inside a workgroup can be told apart."""
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
nw = 80 * 60
st = torch.zeros((nw, 8), dtype=torch.int64, device="cuda")
for i in range(5):
    t.raycast(poses[(7 * i) % 120], depth)
t.synchronize()
L = V.load()
for k in (0, 35, 77):
    assert L.vh_debug_set_raycast_stamps(t._h, st.data_ptr()) == 0
    t.raycast(poses[k], depth)
    t.synchronize()
    L.vh_debug_set_raycast_stamps(t._h, None)
    s = st.cpu().numpy()
    t0 = s[:, 0].min()
    start, end = (s[:, 0] - t0) / 100.0, (s[:, 1] - t0) / 100.0          # microseconds
    life = end - start
    tA, tB = s[:, 4] / 100.0, s[:, 5] / 100.0
    nlist = s[:, 6]
    own, stolen = s[:, 2] & 0xffffffff, s[:, 2] >> 32          # items the wave walked of its own patch / took from its neighbours
    taken, walked = own + stolen, (s[:, 7] & 0xffffffff) + (s[:, 7] >> 32)
    print(f"pose {k}: kernel span {end.max():.1f} us; wave start: median {np.median(start):.1f} max {start.max():.1f}; "
          f"lifetime: mean {life.mean():.1f} median {np.median(life):.1f} p90 {np.percentile(life, 90):.1f} p99 {np.percentile(life, 99):.1f} max {life.max():.1f} us")
    print(f"   own patch: set built after {tA.mean():.2f} us (max {tA.max():.2f}), list final after {tB.mean():.2f} (max {tB.max():.2f}); "
          f"list length mean {nlist.mean():.2f} p99 {np.percentile(nlist, 99):.0f} max {nlist.max()}")
    print(f"   items: own {own.sum()} + stolen {stolen.sum()} of {nlist.sum()} listed; patches with a stolen item {(own < nlist).sum()}; per wave mean {taken.mean():.2f} max {taken.max()}, walked (some ray entered) {walked.sum()}; "
          f"after the list: {((life - tB).sum() / max(1, taken.sum())):.2f} us per item taken")
    g = lambda a: a.reshape(-1, 4)
    gl, gt_, glife, gB = g(nlist).sum(axis=1), g(taken), g(life), g(tB)
    print(f"   groups: items per group mean {gl.mean():.1f} p99 {np.percentile(gl, 99):.0f} max {gl.max()}; heaviest patch of a group mean {g(nlist).max(axis=1).mean():.1f}; "
          f"taken per wave inside a group: spread (max - min) mean {(gt_.max(axis=1) - gt_.min(axis=1)).mean():.2f}; group life (slowest wave) mean {glife.max(axis=1).mean():.1f} max {glife.max(axis=1).max():.1f}; "
          f"slowest list of a group final after mean {gB.max(axis=1).mean():.2f}")
    slow = np.argsort(-life)[:6]
    print("   slowest waves: (life, own list, taken, walked, list final at, group items)",
          [(round(float(life[i]), 1), int(nlist[i]), int(taken[i]), int(walked[i]), round(float(tB[i]), 1), int(gl[i // 4])) for i in slow])
    h, _ = np.histogram(end, bins=10, range=(0, end.max()))
    print("   waves ending per tenth of the span:", h.tolist())
