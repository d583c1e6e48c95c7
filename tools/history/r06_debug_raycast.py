#!/usr/bin/env python3
"""Debug: where does the raycast differ from the oracle?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import oracle as O
import voxelhashing_demo_amd as V
from voxelhashing_demo_amd import synth
W, H = 640, 480
kw = dict(numBuckets=1 << 18, numVoxelBlocks=1 << 14, voxelSize=0.02)
ot = O.OracleTable(O.default_params(**kw), W, H, 1)
gt = V.SDFHashtable(V.default_params(**kw), W, H, 1)
poses, prims = synth.camera_loop(500), synth.room_primitives()
for i in (0, 3, 6, 9, 30, 33):
    v = synth.render_room_verts(poses[i], W, H, prims, device="cuda")
    ot.integrate_mt(poses[i], v.cpu().numpy(), 8)
    gt.integrate(poses[i], v)
gt.synchronize()
d = torch.empty((H, W), dtype=torch.float32, device="cuda")
for beam in (1, 2):
    gt.set_option("raycast_beam", beam)
    for pi in (3, 20):
        od = ot.raycast(poses[pi], 0.1, 5.0)
        for rep in range(3):
            d.fill_(-7.0)
            gt.raycast(poses[pi], d, 0.1, 5.0)
            gt.synchronize()
            g = d.cpu().numpy()
            bad = g.view(np.uint32) != od.view(np.uint32)
            print(f"beam {beam} pose {pi} rep {rep}: mismatches {int(bad.sum())}; unwritten {int((g == -7.0).sum())}; oracle 0 / gpu hit {int((bad & (od == 0) & (g != 0)).sum())}; "
                  f"oracle hit / gpu 0 {int((bad & (od != 0) & (g == 0)).sum())}; both hit {int((bad & (od != 0) & (g != 0)).sum())}")
            if bad.any():
                ys, xs = np.nonzero(bad)
                patches = {}
                for y, x in zip(ys.tolist(), xs.tolist()):
                    patches[(x // 8, y // 8)] = patches.get((x // 8, y // 8), 0) + 1
                pl = sorted(patches.items(), key=lambda kv: -kv[1])
                print("   patches with mismatches:", len(pl), "of 4800; worst", pl[:8])
                k = 0
                for y, x in list(zip(ys.tolist(), xs.tolist()))[:8]:
                    print(f"   ({x},{y}) oracle {od[y, x]:.6f} gpu {g[y, x]:.6f}")
L = V.load()
st = torch.zeros((4800, 8), dtype=torch.int64, device="cuda")
gt.set_option("raycast_beam", 2)
assert L.vh_debug_set_raycast_stamps(gt._h, st.data_ptr()) == 0
gt.raycast(poses[3], d, 0.1, 5.0)
gt.synchronize()
L.vh_debug_set_raycast_stamps(gt._h, None)
s = st.cpu().numpy()
print("stamps: waves with records", int((s[:, 0] != 0).sum()), "sum nList", int(s[:, 6].sum()), "sum taken", int(s[:, 2].sum()), "sum walked", int(s[:, 7].sum()),
      "life us mean", float(((s[:, 1] - s[:, 0]) / 100.0).mean()), "max", float(((s[:, 1] - s[:, 0]) / 100.0).max()))
print("first rows", s[:6].tolist())
