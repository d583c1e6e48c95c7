#!/usr/bin/env python3
"""R bucket-range shards on ONE GPU with the in-process exchange: what a rank's kernels cost per
multi-camera frame when the table is cut R ways (the collectives are not part of this).
   tools/emulate_ranks.py [R=8] [batch=8] [pipeline_shards=1] [per_batch_bins=0]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import voxelhashing_demo_amd as V
from voxelhashing_demo_amd import dist as vdist, synth

W, H = 640, 480
R = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
K = synth.K_matrix(W, H)
kinv = np.linalg.inv(K.astype(np.float64)).astype(np.float32)
prims = synth.room_primitives()
plan = vdist.ShardPlan(1 << 20, R)
PER_BATCH = len(sys.argv) > 4 and int(sys.argv[4]) != 0
# per-frame bins: one record per 16 pixels whatever the number of owners (bench.py's size); per-batch bins: vh_dist's default
cap = max(2048, (-(-W * H // 16) * B * 3 // 2 + R - 1) // R + 1) if PER_BATCH else max(2048, -(-W * H // 16))
shards = [vdist.HipShard(V.default_params(numBuckets=1 << 20, numVoxelBlocks=1 << 16), W, H, 1, plan, r, cap, batch=B,
                         sensor_k_inv=kinv, per_batch_bins=PER_BATCH) for r in range(R)]
if len(sys.argv) > 3:
    for sh in shards:
        sh.table.set_option("pipeline_shards", int(sys.argv[3]))      # 0: two launches per multi-camera frame
if os.environ.get("VH_LEAN") is not None:
    for sh in shards:
        sh.table.set_option("lean_kernels", int(os.environ["VH_LEAN"]))
nf = 64
poses = [synth.camera_loop(500, phase=vdist.camera_phase(r, R))[:nf] for r in range(R)]
depth = [[(synth.render_room_verts(p, W, H, prims, device="cuda")[..., 2] * 5000).round().clamp(0, 65535).to(torch.uint16)
          for p in poses[r]] for r in range(R)]
torch.cuda.synchronize()


def exchange(i):
    ks = [(i * B + b) % nf for b in range(B)]
    vdist.loopback_step(shards, [[poses[r][k] for k in ks] for r in range(R)], [[None] * B for _ in range(R)],
                        [[depth[r][k] for k in ks] for r in range(R)])


for i in range(4):
    exchange(i)
torch.cuda.synchronize()
shards[0].table.set_profiling(True)
n = 6
for i in range(n):
    exchange(4 + i)
torch.cuda.synchronize()
kt = shards[0].table.kernel_times(reset=True)
frames = n * B
print(f"R={R} batch={B}: rank 0 per multi-camera frame ({R} cameras): one-launch frames {1e3*kt['frame_pipelined_ms']/frames:.2f} us "
      f"(B + 1 launches per batch; two-launch form: scan+claim {1e3*kt['frame_scan_claim_ms']/frames:.2f} us, "
      f"commit+integrate {1e3*kt['frame_commit_integrate_ms']/frames:.2f} us); shard {shards[0].table.num_entries*20/1e6:.1f} MB, "
      f"occupied {shards[0].table.counters()['occupied']}, bins {cap*16*R*(1 if PER_BATCH else B)/1e6:.2f} MB "
      f"({'one per (owner, batch)' if PER_BATCH else 'one per (owner, frame)'}, {cap} records) and packets {shards[0].packet_floats*4*R*B/1e6:.1f} MB received per exchange")
fill = max(int(sh.bins_recv[:, :, 0, 0].max().item()) for sh in shards)
print(f"fullest bin of the last exchange: {fill} of {cap - 1} records; bin overflows over the run: {sum(sh.table.counters()['bin_overflow'] for sh in shards)}")
# key generation of a batch, timed on the device
sh = shards[0]
ks = list(range(B))
for _ in range(3):
    sh.generate_all([poses[0][k] for k in ks], [None] * B, [depth[0][k] for k in ks])
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(50):
    sh.generate_all([poses[0][k] for k in ks], [None] * B, [depth[0][k] for k in ks])
torch.cuda.synchronize()
print(f"key generation + packets of a batch of {B}: {1e6*(time.perf_counter()-t)/50:.1f} us")
