#!/usr/bin/env python3
"""A/B of kernel variants in ONE process, interleaved rounds (per-dispatch HIP event timing).

  python tools/ab_kernels.py --option flatten_variant --values 0 1 [--workload C2]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--option", default="flatten_variant")
    ap.add_argument("--values", type=int, nargs="+", default=[0, 1])
    ap.add_argument("--workload", default="C2")
    ap.add_argument("--frames", type=int, default=60)
    ap.add_argument("--rounds", type=int, default=10)
    ap.add_argument("--per-round", type=int, default=50)
    ap.add_argument("--pipeline", type=int, default=0)
    ap.add_argument("--batch", type=int, default=0, help="frames per vh_integrate_batch call (0: vh_integrate)")
    ap.add_argument("--set", nargs="*", default=[], help="name=value options set once")
    ap.add_argument("--preset", nargs="*", default=[], help="name=value options set before the first frame (overflow_list=1 ...)")
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
    verts = [synth.render_room_verts(p, Wd, Ht, prims, device=dev) for p in poses]
    stream = torch.cuda.Stream(device=dev)
    t = V.SDFHashtable(V.default_params(numBuckets=wl["buckets"], numVoxelBlocks=wl["blocks"], voxelSize=wl["voxel"]),
                       Wd, Ht, V.SEM_PINHOLE, stream=stream)
    if wl.get("band"):
        t.set_alloc_band(wl["band"])
    for kv in wl.get("options", []):            # (the workload's own options: C2band's band_mode ...)
        k, v = kv.split("=")
        t.set_option(k, int(v))
    for kv in a.preset:
        k, v = kv.split("=")
        t.set_option(k, int(v))
    for i in range(a.frames):
        t.integrate(poses[i], verts[i])
    t.synchronize()
    t.set_option("pipeline", a.pipeline)
    for kv in a.set:
        k, v = kv.split("=")
        t.set_option(k, int(v))
    res = {v: [] for v in a.values}
    for r in range(a.rounds):
        for v in a.values:
            t.set_option(a.option, v)
            t.set_profiling(True)
            if a.batch:
                for i in range(0, a.per_round, a.batch):
                    ks = [(r * a.per_round + i + j) % a.frames for j in range(a.batch)]
                    t.integrate_batch([poses[k] for k in ks], [verts[k] for k in ks])
            else:
                for i in range(a.per_round):
                    k = (r * a.per_round + i) % a.frames
                    t.integrate(poses[k], verts[k])
            kt = t.kernel_times(reset=True)
            t.set_profiling(False)
            res[v].append({k: 1e3 * kt[k] / kt["launches"] for k in kt if k.endswith("_ms") and k != "raycast_ms"})
    for v in a.values:
        keys = res[v][0].keys()
        print(f"{a.option}={v}: " + "  ".join(
            f"{k[:-3]} med {np.median([x[k] for x in res[v]]):.2f} min {np.min([x[k] for x in res[v]]):.2f} us" for k in keys
            if np.max([x[k] for x in res[v]]) > 0))


if __name__ == "__main__":
    main()
