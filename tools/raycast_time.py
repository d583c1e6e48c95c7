#!/usr/bin/env python3
"""Per-dispatch time of the raycast over bench.py's poses: fuse the first `--frames` frames of a workload, then time
vh_raycast at the poses (7 i) mod frames, i < --poses, with HIP events on the table's stream (vh_set_profiling), `--rounds`
times; prints mean / min / max of the per-pose means.  VOXELHASH_LIB=<.so> times another build on the same box.

  python tools/raycast_time.py [--workload C2|C3] [--option name=value ...] [--normals]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C2")
    ap.add_argument("--frames", type=int, default=120)
    ap.add_argument("--poses", type=int, default=50)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--option", action="append", default=[])
    ap.add_argument("--normals", action="store_true")
    ap.add_argument("--label", default="")
    a = ap.parse_args()
    import torch

    import voxelhashing_demo_amd as V
    from bench import WORKLOADS
    from voxelhashing_demo_amd import synth
    wl = WORKLOADS[a.workload]
    Wd, Ht = wl["width"], wl["height"]
    poses = synth.camera_loop(wl.get("loop", wl["frames"]))[:a.frames]
    prims = synth.room_primitives()
    t = V.SDFHashtable(V.default_params(numBuckets=wl["buckets"], numVoxelBlocks=wl["blocks"], voxelSize=wl["voxel"]), Wd, Ht, V.SEM_PINHOLE)
    for p in poses:
        t.integrate(p, synth.render_room_verts(p, Wd, Ht, prims, device="cuda"))
    t.synchronize()
    for kv in a.option:
        k, v = kv.split("=")
        t.set_option(k, int(v))
    depth = torch.empty((Ht, Wd), dtype=torch.float32, device="cuda")
    nrm = torch.empty((Ht, Wd, 4), dtype=torch.float32, device="cuda")
    ks = [(7 * i) % a.frames for i in range(a.poses)]
    per = np.zeros((a.rounds, len(ks)))
    for k in ks[:5]:
        t.raycast(poses[k], depth)
    t.synchronize()
    t.set_profiling(True)
    for r in range(a.rounds):
        for j, k in enumerate(ks):
            t.kernel_times(reset=True)
            if a.normals:
                t.raycast_normals(poses[k], depth, nrm)
            else:
                t.raycast(poses[k], depth)
            kt = t.kernel_times(reset=True)
            per[r, j] = 1e3 * kt["raycast_ms"] / max(1, kt["raycast_launches"])
    m = per.mean(axis=0)
    print(f"{a.label or os.environ.get('VOXELHASH_LIB', 'lib')} {a.workload} {' '.join(a.option)}{' normals' if a.normals else ''}: "
          f"mean {m.mean():.2f} us  min pose {m.min():.2f}  max pose {m.max():.2f}  hits {float((depth > 0).float().mean()):.3f}")


if __name__ == "__main__":
    main()
