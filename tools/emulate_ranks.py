#!/usr/bin/env python3
"""R ranks of the NATIVE exchange (vh_dist_step_batch -- the code bench.py --gpus N runs) on ONE GPU, joined by the library's
loop-back transport, one host thread per rank: what a rank's launches cost per multi-camera frame when the table is cut R
ways, how full the key bins get, and what the whole rig sustains.  The bytes cross by hipMemcpyAsync instead of xGMI, and
the R ranks share one GPU, so the frames/s figure is a rig number, not a scaling point.
   tools/emulate_ranks.py [R=8] [batch=8] [workload=C2|C5] [steps=12] [fused_generation=default|0|1]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import voxelhashing_demo_amd as V
from voxelhashing_demo_amd import dist as vdist
from voxelhashing_demo_amd import synth

R = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
WL = sys.argv[3] if len(sys.argv) > 3 else "C2"
STEPS = int(sys.argv[4]) if len(sys.argv) > 4 else 12
FUSED = int(sys.argv[5]) if len(sys.argv) > 5 else None
W, H, NB, VOX, BLOCKS = (640, 480, 1 << 20, 0.02, 1 << 16) if WL == "C2" else (1920, 1080, 1 << 24, 0.01, 1 << 16)
kinv = np.linalg.inv(synth.K_matrix(W, H).astype(np.float64)).astype(np.float32)
prims = synth.room_primitives()
params = V.default_params(numBuckets=NB, numVoxelBlocks=BLOCKS, voxelSize=VOX)
g = vdist.NativeGroup(params, W, H, 1, R, B, sensor_k_inv=kinv)
if FUSED is not None:
    for nd in g.ranks:
        nd.set_option("fused_generation", FUSED)
nf = 32 if WL == "C2" else 8
poses = [synth.camera_loop(500, phase=vdist.camera_phase(r, R))[:nf] for r in range(R)]
depth = [[(synth.render_room_verts(p, W, H, prims, device="cuda")[..., 2] * 5000).round().clamp(0, 65535).to(torch.uint16)
          for p in poses[r]] for r in range(R)]
torch.cuda.synchronize()


def exchange(i):
    ks = [(i * B + b) % nf for b in range(B)]
    g.step([[poses[r][k] for k in ks] for r in range(R)], [[depth[r][k] for k in ks] for r in range(R)])


for i in range(4):
    exchange(i)
g.flush()
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(STEPS):
    exchange(4 + i)
g.flush()
torch.cuda.synchronize()
wall = time.perf_counter() - t0
for t in g.tables:
    t.set_profiling(True)
PROF = 10            # (behind a flush the first two exchanges generate with launches of their own: 8 of 10 carry the role)
for i in range(PROF):
    exchange(4 + STEPS + i)
g.flush()
torch.cuda.synchronize()
frames = PROF * B
cap = max(8192, (-(-W * H // 16) * B * 3 // 2 + R - 1) // R + 1)
per_rank = []
for r, t in enumerate(g.tables):
    kt = t.kernel_times(reset=True)
    per_rank.append(1e3 * kt["frame_pipelined_ms"] / frames)
    t.set_profiling(False)
c0 = g.tables[0].counters()
print(f"{WL} R={R} batch={B} transport={g.ranks[0].transport}: frame_multi_pipelined_kernel per multi-camera frame ({R} cameras), per rank: "
      + " ".join(f"{x:.2f}" for x in per_rank) + f" us; shard {g.tables[0].num_entries * 20 / 1e6:.1f} MB, rank 0 occupied {c0['occupied']}")
print(f"key bins: {cap} records per (owner, batch) = {cap * 16 * R / 1e6:.2f} MB per rank and exchange; bin overflows over the run: "
      f"{sum(t.counters()['bin_overflow'] for t in g.tables)}; host time in vh_dist_step_batch per exchange (rank 0): "
      f"{1e6 * g.ranks[0].host_stats()[0] / g.ranks[0].host_stats()[1]:.0f} us")
print(f"rig throughput (all {R} ranks on this one GPU, loop-back copies): {STEPS * B * R / wall:.0f} frames/s "
      f"= {1e6 * wall / (STEPS * B):.1f} us per multi-camera frame for the {R} shards together")
g.close()
