#!/usr/bin/env python3
"""Frames fed from HOST memory every frame (pinned buffers, copy stream + table stream, two device
buffers): float4 vertex maps (4.9 MB/frame) against uint16 sensor images (0.6 MB/frame, vh_integrate_depth)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import voxelhashing_demo_amd as V
from voxelhashing_demo_amd import synth

W, H, nf, steps = 640, 480, 64, 1000
poses = synth.camera_loop(500)[:nf]
prims = synth.room_primitives()
kinv = np.linalg.inv(synth.K_matrix(W, H).astype(np.float64)).astype(np.float32)
dverts = [synth.render_room_verts(p, W, H, prims, device="cuda") for p in poses]
h_verts = [v.cpu().pin_memory() for v in dverts]
h_depth = [(v[..., 2] * 5000).round().clamp(0, 65535).to(torch.uint16).cpu().pin_memory() for v in dverts]
del dverts
torch.cuda.synchronize()
table_stream, copy_stream = torch.cuda.Stream(), torch.cuda.Stream()
for name, host, fn in (("float4 vertex maps", h_verts, "integrate"), ("uint16 sensor images", h_depth, "integrate_depth")):
    t = V.SDFHashtable(V.default_params(numBuckets=1 << 20, numVoxelBlocks=1 << 18), W, H, V.SEM_PINHOLE, stream=table_stream)
    dev = [torch.empty_like(host[0], device="cuda") for _ in range(2)]
    copied = [torch.cuda.Event() for _ in range(2)]
    used = [torch.cuda.Event() for _ in range(2)]

    def run(n):
        for i in range(n):
            s, k = i & 1, i % nf
            with torch.cuda.stream(copy_stream):
                if i >= 2:
                    copy_stream.wait_event(used[s])
                dev[s].copy_(host[k], non_blocking=True)
                copied[s].record(copy_stream)
            with torch.cuda.stream(table_stream):
                table_stream.wait_event(copied[s])
                if fn == "integrate":
                    t.integrate(poses[k], dev[s])
                else:
                    t.integrate_depth(poses[k], dev[s], kinv)
                used[s].record(table_stream)
        torch.cuda.synchronize()

    run(200)
    t0 = time.perf_counter()
    run(steps)
    dt = time.perf_counter() - t0
    mb = host[0].numel() * host[0].element_size() / 1e6
    print(f"{name}: {steps / dt:.0f} frames/s fed over PCIe ({mb:.2f} MB/frame, {mb * steps / dt / 1e3:.1f} GB/s)")
    t.close()
