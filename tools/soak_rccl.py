#!/usr/bin/env python3
"""Longer runs of the native exchange on RCCL with R > 1 ranks on one GPU (tests/rccl_rank.py, one process per rank, every
rank a host of its own to RCCL): many exchanges, so that the three buffer sets, the deferred frame and the event hand-offs
between the library's streams and RCCL's kernels are exercised well beyond the tests' four exchanges.  Every rank compares its
shard with its slice of ONE oracle table and its raycast with the oracle's."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
from test_gpu_dist_rccl import run_ranks  # noqa: E402

CASES = [(4, 4, True, ["exchanges=40"]), (8, 2, True, ["exchanges=30", "buckets=16", "blocks=12"]), (2, 8, False, ["exchanges=25", "size=640x480", "buckets=20", "blocks=15"]),
         (3, 3, True, ["exchanges=30"])]
for world, batch, sensor, extra in CASES:
    t0 = time.time()
    run_ranks(world, batch, sensor, *extra, timeout=1500)
    print(f"ranks {world} x batch {batch} x {extra[0]} ({'uint16' if sensor else 'vertex maps'}, {' '.join(extra[1:]) or '320x240'}): every shard and view equal to the oracle's, {time.time() - t0:.0f} s", flush=True)
