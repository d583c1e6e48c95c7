#!/usr/bin/env python3
"""Inputs of the scaling model (tools/scaling_model.py -> profiles/r05_scaling_model.json), measured on ONE GPU, one rank at
a time: R bucket-range shards with the in-process exchange (dist.loopback_step: every rank's kernels run one after the
other on the one GPU, nothing is concurrent), rank 0's launches timed per dispatch.  For R = 1, 2, 4, 8:
  frame_us        frame_multi_pipelined_kernel per multi-camera frame on rank 0 (R cameras, the shard = 1/R of the table)
  frame_index_us  the same with the walk-free frame (flatten_variant 4)
  gen_us          key generation + packet of one batch of this rank's camera
  bytes on the wire per rank and exchange (key bins at the native exchange's default capacity, sensor packets)
   tools/scaling_inputs.py [workload=C2|C5] [batch=8] > gpurun_out/r05/scaling_inputs_<workload>.json"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import voxelhashing_demo_amd as V
from voxelhashing_demo_amd import dist as vdist
from voxelhashing_demo_amd import synth

WL = sys.argv[1] if len(sys.argv) > 1 else "C2"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
W, H, NB, VOX, BLOCKS = (640, 480, 1 << 20, 0.02, 1 << 16) if WL == "C2" else (1920, 1080, 1 << 24, 0.01, 1 << 16)
kinv = np.linalg.inv(synth.K_matrix(W, H).astype(np.float64)).astype(np.float32)
prims = synth.room_primitives()
nf = 32 if WL == "C2" else 8
out = dict(workload=WL, width=W, height=H, buckets=NB, batch=B, ranks={})
for R in (1, 2, 4, 8):
    plan = vdist.ShardPlan(NB, R)
    cap = max(2048, -(-W * H // 16))                 # (per-frame bins of the Python exchange: room for a whole frame's keys)
    shards = [vdist.HipShard(V.default_params(numBuckets=NB, numVoxelBlocks=BLOCKS, voxelSize=VOX), W, H, 1, plan, r, cap, batch=B,
                             sensor_k_inv=kinv) for r in range(R)]
    poses = [synth.camera_loop(500, phase=vdist.camera_phase(r, R))[:nf] for r in range(R)]
    depth = [[(synth.render_room_verts(p, W, H, prims, device="cuda")[..., 2] * 5000).round().clamp(0, 65535).to(torch.uint16)
              for p in poses[r]] for r in range(R)]
    torch.cuda.synchronize()

    def exchange(i):
        ks = [(i * B + b) % nf for b in range(B)]
        vdist.loopback_step(shards, [[poses[r][k] for k in ks] for r in range(R)], [[None] * B for _ in range(R)],
                            [[depth[r][k] for k in ks] for r in range(R)])

    rec = {}
    for label, variant in (("frame_us", 3), ("frame_index_us", 4)):
        for sh in shards:
            sh.table.set_option("flatten_variant", variant)
        for i in range(4):
            exchange(i)
        torch.cuda.synchronize()
        shards[0].table.set_profiling(True)
        n = 5
        for i in range(n):
            exchange(4 + i)
        torch.cuda.synchronize()
        kt = shards[0].table.kernel_times(reset=True)
        shards[0].table.set_profiling(False)
        one = kt["frame_pipelined_ms"] > 0
        rec[label] = round(1e3 * (kt["frame_pipelined_ms"] if one else kt["frame_scan_claim_ms"] + kt["frame_commit_integrate_ms"]) / (n * B), 2)
        rec["launches_per_frame"] = 1 if one else 2
    sh = shards[0]
    ks = list(range(B))
    for _ in range(3):
        sh.generate_all([poses[0][k] for k in ks], [None] * B, [depth[0][k] for k in ks])
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(30):
        sh.generate_all([poses[0][k] for k in ks], [None] * B, [depth[0][k] for k in ks])
    torch.cuda.synchronize()
    rec["gen_us"] = round(1e6 * (time.perf_counter() - t) / 30, 1)
    c = sh.table.counters()
    rec["shard_mb"] = round(sh.table.num_entries * 20 / 1e6, 1)
    rec["occupied_rank0"] = c["occupied"]
    native_cap = max(8192, (-(-W * H // 16) * B * 3 // 2 + R - 1) // R + 1)        # vh_dist_create's default bin
    rec["bin_bytes_per_peer"] = native_cap * 16
    rec["packet_bytes_per_peer"] = 4 * (36 + W * H // 2) * B
    out["ranks"][str(R)] = rec
    for s in shards:
        s.table.close()
    del shards, depth
    torch.cuda.empty_cache()
print(json.dumps(out))
