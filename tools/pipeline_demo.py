#!/usr/bin/env python3
"""The whole reconstruction loop of the demo on synthetic sensor data, every stage on the GPU
through the C-ABI: uint16 depth -> vertex/normal maps (preProcess) -> pose from frame-to-model ICP
against a raycast of the model (CameraTracking::Align; bypassed in the reference, Application.cpp:75)
-> TSDF integration (SDF_Hashtable::integrate) -> periodic garbage collection.  Prints the time per
stage and the drift against the true trajectory.   tools/pipeline_demo.py [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import voxelhashing_demo_amd as V
from voxelhashing_demo_amd import synth, tracking

W, H = 640, 480
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
gt = synth.camera_loop(500)[200:200 + N]
prims = synth.room_primitives()
K = synth.K_matrix(W, H)
kinv = np.linalg.inv(K.astype(np.float64)).astype(np.float32)
# synthetic sensor frames: depth in 1/5000 m, as the TUM sequences the demo reads (Application.cpp:38-42)
depth16 = [(synth.render_room_verts(p, W, H, prims, device="cuda")[..., 2] * 5000.0).round().clamp(0, 65535).to(torch.uint16)
           for p in gt]
torch.cuda.synchronize()           # the sensor frames were made on torch's default stream
stream = torch.cuda.Stream()
table = V.SDFHashtable(V.default_params(numBuckets=1 << 20, numVoxelBlocks=1 << 16), W, H, V.SEM_PINHOLE, stream=stream)
trk = tracking.CameraTracking(W, H, K, stream=stream, flags=tracking.ICP_ABS_DISTANCE | tracking.ICP_NEED_TARGET)
verts, normals = torch.empty((H, W, 4), device="cuda"), torch.empty((H, W, 4), device="cuda")
tp, tn = torch.empty_like(verts), torch.empty_like(verts)
ray = torch.empty((H, W), device="cuda")
stage = dict(preprocess=0.0, raycast_target=0.0, align=0.0, integrate=0.0, collect=0.0)


def timed(name, fn):
    stream.synchronize()
    t = time.perf_counter()
    fn()
    stream.synchronize()
    stage[name] += time.perf_counter() - t


pose = np.asarray(gt[0], np.float64).reshape(4, 4)
errs = []
with torch.cuda.stream(stream):
    for k in range(N):
        timed("preprocess", lambda: V.preprocess(depth16[k], kinv, verts, normals, stream=stream))
        if k > 0:
            def target():
                table.raycast(pose.astype(np.float32), ray)
                tracking.depth_to_maps(ray, kinv, tp, tn, stream=stream)
            timed("raycast_target", target)
            delta = [None]
            timed("align", lambda: delta.__setitem__(0, trk.Align(verts, tp, tn)))
            pose = pose @ delta[0].astype(np.float64)
        timed("integrate", lambda: table.integrate(pose.astype(np.float32), verts))
        if k % 20 == 19:
            timed("collect", lambda: table.garbage_collect(0.5))
        errs.append(float(np.abs(pose[:3, 3] - np.asarray(gt[k], np.float64).reshape(4, 4)[:3, 3]).max()))
travel = float(np.linalg.norm(np.asarray(gt[-1], np.float64).reshape(4, 4)[:3, 3] - np.asarray(gt[0], np.float64).reshape(4, 4)[:3, 3]))
print(f"{N} frames, {travel:.2f} m between first and last camera, blocks {table.counters()['allocated_total']}")
for name, t in stage.items():
    n = N - 1 if name in ("raycast_target", "align") else (N // 20 if name == "collect" else N)
    print(f"  {name:15s} {1e6 * t / max(1, n):8.1f} us per call (host-timed, synchronised)")
print(f"  drift: max {1e3 * max(errs):.2f} mm, final {1e3 * errs[-1]:.2f} mm")
