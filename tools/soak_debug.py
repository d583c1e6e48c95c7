#!/usr/bin/env python3
"""Lockstep GPU / oracle run; reports the first frame after which a block differs.
   soak_debug.py <frames> <gc 0|1> <raycast 0|1> <check every>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import oracle as O
import voxelhashing_demo_amd as V
from voxelhashing_demo_amd import synth

W, H = 640, 480
N, GC, RC, EVERY = (int(x) for x in sys.argv[1:5])
poses = synth.camera_loop(500)[:N]
prims = synth.room_primitives()
kw = dict(numBuckets=1 << 20, numVoxelBlocks=1 << 15)
t = V.SDFHashtable(V.default_params(**kw), W, H, V.SEM_PINHOLE)
ot = O.OracleTable(O.default_params(**kw), W, H, O.SEM_PINHOLE)
depth = torch.empty((H, W), device="cuda")
for i, p in enumerate(poses):
    v = synth.render_room_verts(p, W, H, prims, device="cuda")
    t.integrate(p, v)
    ot.integrate(p, v.cpu().numpy())
    if RC and i % 10 == 9:
        t.raycast(p, depth)
    if GC and i % 50 == 49:
        t.garbage_collect(0.5)
        ot.garbage_collect(0.5)
    if i % EVERY == EVERY - 1 or i == N - 1:
        t.synchronize()
        ref, tab = ot.hash_table(), t.hash_table()
        same_tab = np.array_equal(tab["pos"], ref["pos"]) and np.array_equal(tab["ptr"] != -1, ref["ptr"] != -1)
        ovol, vol = ot.sdf_blocks(), t.sdf_blocks()
        bad = []
        for j in np.nonzero(ref["ptr"] != -1)[0]:
            go, gg = int(ref["ptr"][j]), int(tab["ptr"][j])
            a, b = ovol[go:go + 512], vol[gg:gg + 512]
            if not np.array_equal(a.view(np.uint32), b.view(np.uint32)):
                d = np.nonzero((a["sdf"] != b["sdf"]) | (a["weight"] != b["weight"]))[0]
                bad.append((int(j), tuple(ref["pos"][j].tolist()), len(d), float(np.abs(a["sdf"] - b["sdf"]).max()),
                            float(np.abs(a["weight"] - b["weight"]).max())))
        print("frame", i, "table same", same_tab, "blocks", int((ref["ptr"] != -1).sum()), "bad blocks", len(bad), bad[:4], flush=True)
        if bad or not same_tab:
            break
