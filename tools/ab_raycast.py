#!/usr/bin/env python3
"""A/B of raycast variants in ONE process, interleaved rounds (per-dispatch HIP event timing).

  python tools/ab_raycast.py --option raycast_patch --values 0 1
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--option", default="raycast_patch")
    ap.add_argument("--values", type=int, nargs="+", default=[0, 1])
    ap.add_argument("--workload", default="C2")
    ap.add_argument("--frames", type=int, default=120)
    ap.add_argument("--rounds", type=int, default=8)
    ap.add_argument("--per-round", type=int, default=20)
    a = ap.parse_args()
    import torch

    import voxelhashing_demo_amd as V
    from bench import WORKLOADS
    from voxelhashing_demo_amd import synth
    wl = WORKLOADS[a.workload]
    Wd, Ht = wl["width"], wl["height"]
    dev = torch.device("cuda", 0)
    poses = synth.camera_loop(wl["frames"])[:a.frames]
    prims = synth.room_primitives()
    stream = torch.cuda.Stream(device=dev)
    t = V.SDFHashtable(V.default_params(numBuckets=wl["buckets"], numVoxelBlocks=wl["blocks"], voxelSize=wl["voxel"]),
                       Wd, Ht, V.SEM_PINHOLE, stream=stream)
    for i in range(a.frames):
        t.integrate(poses[i], synth.render_room_verts(poses[i], Wd, Ht, prims, device=dev))
    t.synchronize()
    depth = torch.empty((Ht, Wd), dtype=torch.float32, device=dev)
    res = {v: [] for v in a.values}
    for r in range(a.rounds):
        for v in a.values:
            t.set_option(a.option, v)
            t.set_profiling(True)
            for i in range(a.per_round):
                t.raycast(poses[(7 * (r * a.per_round + i)) % a.frames], depth)
            kt = t.kernel_times(reset=True)
            t.set_profiling(False)
            res[v].append(1e3 * kt["raycast_ms"] / kt["raycast_launches"])
    for v in a.values:
        us = np.median(res[v])
        print(f"{a.option}={v}: raycast med {us:.1f} us  min {np.min(res[v]):.1f} us  "
              f"= {Wd * Ht / us:.0f} Mpix/s (kernel time)")


if __name__ == "__main__":
    main()
