#!/usr/bin/env python3
"""Where a claim tile's time goes (diagnostics build: tools/ab_variants.sh build stamps -DVH_CLAIM_STAMPS, then
VOXELHASH_LIB=voxelhashing_demo_amd/lib/alt/v_stamps.so python tools/claim_stamps.py [C2band|C2]): per 16x16 claim tile of one
pipelined launch the time to the vertex, through the sample loop, through the drain of the key queue."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import voxelhashing_demo_amd as V
from bench import WORKLOADS
from voxelhashing_demo_amd import synth, _lib
wl = WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "C2band"]
Wd, Ht = wl["width"], wl["height"]
poses = synth.camera_loop(500)[:60]
prims = synth.room_primitives()
verts = [synth.render_room_verts(p, Wd, Ht, prims, device="cuda") for p in poses]
t = V.SDFHashtable(V.default_params(numBuckets=wl["buckets"], numVoxelBlocks=wl["blocks"], voxelSize=wl["voxel"]), Wd, Ht, V.SEM_PINHOLE)
if wl.get("band"):
    t.set_alloc_band(wl["band"])
for kv in sys.argv[2:]:            # name=value options (band_mode=2 ...)
    k, v = kv.split("=")
    t.set_option(k, int(v))
for i in range(60):
    t.integrate(poses[i], verts[i])
t.synchronize()
L = C.CDLL(_lib.LIB_PATH)
tiles = (Wd // 16) * (Ht // 16)
st = torch.zeros((tiles, 4), dtype=torch.int64, device="cuda")
for rep in range(3):
    t.integrate_batch([poses[k] for k in range(8)], [verts[k] for k in range(8)])
t.synchronize()
assert L.vh_debug_set_claim_stamps(C.c_void_p(st.data_ptr())) == 0
t.integrate_batch([poses[20]], [verts[20]])
t.synchronize()
L.vh_debug_set_claim_stamps(None)
s = st.cpu().numpy().astype(np.float64) / 100.0
t0 = s[:, 0].min()
a, b, c, d = s[:, 0] - t0, s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2]
print(f"claim tiles: {tiles}; start spread: median {np.median(a):.1f} p90 {np.percentile(a, 90):.1f} max {a.max():.1f} us")
for name, x in (("vertex load", b), ("sample loop", c), ("drain (probe + claim)", d), ("whole tile", s[:, 3] - s[:, 0])):
    print(f"  {name:24s} mean {x.mean():6.2f}  median {np.median(x):6.2f}  p90 {np.percentile(x, 90):6.2f}  max {x.max():6.2f} us")
print(f"  last tile ends at {(s[:, 3] - t0).max():.1f} us")
