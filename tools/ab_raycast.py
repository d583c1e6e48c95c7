#!/usr/bin/env python3
"""A/B of a raycast / silhouette option in ONE process, interleaved rounds (per-dispatch HIP event timing).

  python tools/ab_raycast.py --option raycast_patch --values 0 1 [--workload C2]
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
    ap.add_argument("--per-round", type=int, default=40)
    ap.add_argument("--blocks", action="store_true", help="time vh_render_blocks instead of vh_raycast")
    a = ap.parse_args()
    import torch

    import voxelhashing_demo_amd as V
    from bench import WORKLOADS
    from voxelhashing_demo_amd import synth
    wl = WORKLOADS[a.workload]
    Wd, Ht = wl["width"], wl["height"]
    dev = torch.device("cuda", 0)
    poses = synth.camera_loop(wl.get("loop", wl["frames"]))[:a.frames]
    prims = synth.room_primitives()
    t = V.SDFHashtable(V.default_params(numBuckets=wl["buckets"], numVoxelBlocks=wl["blocks"], voxelSize=wl["voxel"]),
                       Wd, Ht, V.SEM_PINHOLE)
    for p in poses:
        t.integrate(p, synth.render_room_verts(p, Wd, Ht, prims, device=dev))
    t.synchronize()
    out, out2 = torch.empty((Ht, Wd), device=dev), torch.empty((Ht, Wd), device=dev)
    res = {v: [] for v in a.values}
    key = "render_blocks_ms" if a.blocks else "raycast_ms"
    for r in range(a.rounds):
        for v in a.values:
            t.set_option(a.option, v)
            t.set_profiling(True)
            for i in range(a.per_round):
                pose = poses[(7 * (r * a.per_round + i)) % a.frames]
                if a.blocks:
                    t.render_blocks(pose, out, out2, 0.1, 5.0)
                else:
                    t.raycast(pose, out)
            kt = t.kernel_times(reset=True)
            t.set_profiling(False)
            res[v].append(1e3 * kt[key] / a.per_round)
    for v in a.values:
        print(f"{a.option}={v}: {key[:-3]} med {np.median(res[v]):.2f} min {np.min(res[v]):.2f} us per call")


if __name__ == "__main__":
    main()
