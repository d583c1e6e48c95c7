#!/usr/bin/env python3
"""Two room frames, N calls of vh_icp_align (for rocprofv3 --kernel-trace of the ICP kernels)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from voxelhashing_demo_amd import synth, tracking
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
W, H = (int(a) for a in os.environ.get("ICP_SIZE", "640x480").split("x"))      # ICP_SIZE=1280x960
poses = synth.camera_loop(250)
prims = synth.room_primitives()
K = synth.K_matrix(W, H)
kinv = np.linalg.inv(K.astype(np.float64)).astype(np.float32)
v0 = synth.render_room_verts(poses[100], W, H, prims, device="cuda")
v1 = synth.render_room_verts(poses[101], W, H, prims, device="cuda")
tp, tn = torch.empty_like(v0), torch.empty_like(v0)
tracking.depth_to_maps(v0[..., 2].contiguous(), kinv, tp, tn)
trk = tracking.CameraTracking(W, H, K, flags=3)
if len(sys.argv) > 2 and sys.argv[2] == "step":       # the step API: one round per call, no device-side solve
    for i in range(20 * n):
        trk.build_system(v1, tp, tn, np.eye(4))
    sys.exit(0)
trk.Align(v1, tp, tn)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(n):
    d = trk.Align(v1, tp, tn)
torch.cuda.synchronize()
print("us per align", 1e6 * (time.perf_counter() - t0) / n, "rounds", trk.iterations, "pairs", trk.last[3])
true = np.linalg.inv(np.asarray(poses[100], np.float64).reshape(4, 4)) @ np.asarray(poses[101], np.float64).reshape(4, 4)
print("translation error", np.abs(d[:3, 3] - true[:3, 3]).max())
