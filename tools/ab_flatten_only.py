#!/usr/bin/env python3
"""Walk-only A/B: vh_flatten alone (no claim phase) for each walk variant, one process."""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C3")
    ap.add_argument("--values", type=int, nargs="+", default=[3, 4, 5])
    ap.add_argument("--frames", type=int, default=40)
    a = ap.parse_args()
    import torch
    import voxelhashing_demo_amd as V
    from bench import WORKLOADS
    from voxelhashing_demo_amd import synth
    wl = WORKLOADS[a.workload]
    Wd, Ht = wl["width"], wl["height"]
    poses = synth.camera_loop(wl["frames"])[:a.frames]
    prims = synth.room_primitives()
    t = V.SDFHashtable(V.default_params(numBuckets=wl["buckets"], numVoxelBlocks=wl["blocks"], voxelSize=wl["voxel"]),
                       Wd, Ht, V.SEM_PINHOLE, stream=torch.cuda.Stream())
    for i in range(a.frames):
        t.integrate(poses[i], synth.render_room_verts(poses[i], Wd, Ht, prims, device="cuda"))
    t.synchronize()
    res = {v: [] for v in a.values}
    for r in range(8):
        for v in a.values:
            t.set_option("flatten_variant", v)
            t.set_profiling(True)
            for i in range(20):
                t.flatten(sync=False)
            kt = t.kernel_times(reset=True)
            t.set_profiling(False)
            res[v].append(1e3 * kt["flatten_ms"] / 20)
    n = t.num_entries
    for v in a.values:
        us = float(np.median(res[v]))
        print(f"flatten_variant={v}: {us:.2f} us  ({20 * n / us / 1e6:.2f} TB/s of 20*N bytes), occupied {t.counters()['occupied']}")


if __name__ == "__main__":
    main()
