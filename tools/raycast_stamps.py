#!/usr/bin/env python3
"""Per-wave timeline of one DDA raycast launch on C2 (vh_debug_set_raycast_stamps): when each wave started and ended
on the 100 MHz constant clock, so that dispatch ramp, mean wave lifetime and the tail can be told apart."""
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
nw = 40 * 30 * 4
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
    print(f"pose {k}: kernel span {end.max():.1f} us; wave start: median {np.median(start):.1f} p90 {np.percentile(start, 90):.1f} max {start.max():.1f}; "
          f"lifetime: mean {life.mean():.1f} median {np.median(life):.1f} p90 {np.percentile(life, 90):.1f} p99 {np.percentile(life, 99):.1f} max {life.max():.1f} us")
    order = np.argsort(-life)[:6]
    rounds = (s[:, 2] >> 32).astype(np.float64)
    print("   slowest waves (life us, start us, steps of lane 0, loop rounds of the wave, patch x,y):",
          [(round(float(life[i]), 1), round(float(start[i]), 1), int(s[i, 2] & 0xffffffff), int(rounds[i]), int(s[i, 3] & 0xffff), int((s[i, 3] >> 16) & 0xffff)) for i in order])
    front = (s[:, 3] >> 32) / 100.0
    print(f"   prologue + beam front end per wave: mean {front.mean():.2f} p99 {np.percentile(front, 99):.2f} max {front.max():.2f} us")
    if s[:, 6].max() > 0:
        tA, tB = s[:, 4] / 100.0, s[:, 5] / 100.0
        print(f"   cooperative form: set built after {tA.mean():.2f} us (max {tA.max():.2f}), list resolved after {tB.mean():.2f} (max {tB.max():.2f}); "
              f"list length mean {s[:, 6].mean():.1f} p99 {np.percentile(s[:, 6], 99):.0f} max {s[:, 6].max()}; blocks walked by some lane: mean {s[:, 7].mean():.1f} max {s[:, 7].max()}")
        early, late = start < 2.0, start > 6.0
        for name, m in (("first round of waves", early), ("waves started later", late)):
            if m.any():
                print(f"      {name} ({int(m.sum())}): set built after {tA[m].mean():.2f}, list after {tB[m].mean():.2f}, life {life[m].mean():.2f} us; "
                      f"list {s[m, 6].mean():.1f}; after the list: {((life[m] - tB[m]) / np.maximum(1, s[m, 6])).mean():.2f} us per listed block")
        print(f'      ray set-up done after {((s[:, 2] & 0xffffffff) / 100.0).mean():.2f} us')
        # what sharing the listed blocks inside a 16x16 tile (4 waves) could give: a tile as long as its waves' mean, not their max
        t4 = life.reshape(-1, 4); f4 = tB.reshape(-1, 4)
        walk4 = (t4 - f4)
        shared = f4.max(axis=1) + walk4.mean(axis=1)
        print(f"      tiles: slowest wave per tile mean {t4.max(axis=1).mean():.1f} max {t4.max(axis=1).max():.1f} us; with the walks of a tile shared evenly: "
              f"mean {shared.mean():.1f} max {shared.max():.1f} us; the ten slowest tiles now {np.sort(t4.max(axis=1))[-10:].round(1).tolist()} shared {np.sort(shared)[-10:].round(1).tolist()}")
        slow = np.argsort(-life)[:6]
        print("   slowest waves: (life, list, walked)", [(round(float(life[i]), 1), int(s[i, 6]), int(s[i, 7])) for i in slow])
    ok = rounds > 0
    if not ok.any():
        continue
    print(f"   loop rounds per wave: mean {rounds[ok].mean():.1f} p99 {np.percentile(rounds[ok], 99):.0f} max {rounds.max():.0f}; "
          f"us per round: mean wave {((life[ok] - front[ok]) / rounds[ok]).mean():.2f}, slowest waves {np.mean([(life[i] - front[i]) / max(1, rounds[i]) for i in order]):.2f}")
    h, _ = np.histogram(end, bins=10, range=(0, end.max()))
    print("   waves ending per tenth of the span:", h.tolist())
