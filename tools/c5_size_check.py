#!/usr/bin/env python3
"""One-off: the C5 per-stream size (1920x1080, 2^24 buckets = 1.68 GB table, 2^21 blocks = 8.6 GB volume,
1 cm voxels) on one GPU against the oracle: index widths, grid limits, memory."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import oracle as O
import voxelhashing_demo_amd as V
from voxelhashing_demo_amd import synth

W, H = 1920, 1080
kw = dict(numBuckets=1 << 24, numVoxelBlocks=1 << 21, voxelSize=0.01)
poses = synth.camera_loop(500)[::40][:4]
prims = synth.room_primitives()
gt = V.SDFHashtable(V.default_params(**kw), W, H, V.SEM_PINHOLE)
ot = O.OracleTable(O.default_params(**kw), W, H, O.SEM_PINHOLE)
for i, p in enumerate(poses):
    v = synth.render_room_verts(p, W, H, prims, device="cuda")
    t0 = time.perf_counter()
    gt.integrate(p, v)
    gt.synchronize()
    t1 = time.perf_counter()
    ot.integrate_mt(p, v.cpu().numpy(), 16)
    print(f"frame {i}: gpu {1e3*(t1-t0):.2f} ms, oracle {time.perf_counter()-t1:.1f} s, occupied {gt.counters()['occupied']}", flush=True)
a, b = ot.hash_table(), gt.hash_table()
assert np.array_equal(a["pos"], b["pos"]) and np.array_equal(a["ptr"] != -1, b["ptr"] != -1)
live = np.nonzero(a["ptr"] != -1)[0]
ov, gv = ot.sdf_blocks(), gt.sdf_blocks()
for i in live[:: max(1, len(live) // 3000)]:
    assert np.array_equal(ov[int(a["ptr"][i]):int(a["ptr"][i]) + 512].view(np.uint32), gv[int(b["ptr"][i]):int(b["ptr"][i]) + 512].view(np.uint32))
depth = torch.empty((H, W), device="cuda")
gt.raycast(poses[-1], depth)
torch.cuda.synchronize()
ref = ot.raycast(poses[-1])
assert np.array_equal(depth.cpu().numpy().view(np.uint32), ref.view(np.uint32))
gt.set_profiling(True)
for i in range(20):
    gt.integrate(poses[i % 4], synth.render_room_verts(poses[i % 4], W, H, prims, device="cuda"))
kt = gt.kernel_times(reset=True)
print(f"C5-SIZE OK: {len(live)} blocks, raycast hits {(ref>0).mean():.2f}; launch 1 {1e3*kt['frame_scan_claim_ms']/kt['launches']:.1f} us "
      f"({(20*84e6+16*W*H)/ (1e3*kt['frame_scan_claim_ms']/kt['launches']*1e-6)/1e12:.2f} TB/s), launch 2 {1e3*kt['frame_commit_integrate_ms']/kt['launches']:.1f} us")
