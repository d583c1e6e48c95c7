#!/usr/bin/env python3
"""Condense rocprofv3 CSV output into the small summaries kept under profiles/.

  prof_summary.py stats  <rocprof_dir> <out.csv>      kernel_stats.csv rows of this repo's kernels
                                                       (+ per-kernel average from the trace)
  prof_summary.py pmc    <rocprof_dir> <out.json> [skip]   per-kernel mean of every collected counter,
                                                       skipping the first `skip` dispatches of each kernel
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

def short(name):
    """Name of one of this repo's kernels (they all live in namespace vh) without namespace and
    arguments, template arguments kept (frame_scan_claim_kernel<3> = ballot walk, <4> = occupancy-index
    walk, ...); None for anything else (torch, rocclr copies)."""
    i = name.find("vh::")
    if i < 0:
        return None
    rest = name[i + 4:]
    depth, end = 0, len(rest)
    for j, ch in enumerate(rest):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            end = j
            break
    return rest[:end].strip()


def find(d, pat):
    hits = glob.glob(os.path.join(d, "**", pat), recursive=True)
    if not hits:
        raise SystemExit(f"no {pat} under {d}")
    return hits[0]


def stats(d, out):
    rows = list(csv.DictReader(open(find(d, "*kernel_stats.csv"))))
    keep = [r for r in rows if short(r.get("Name", ""))]
    other_ns = sum(float(r["TotalDurationNs"]) for r in rows if not short(r.get("Name", "")))
    with open(out, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        for r in keep:
            w.writerow(r)
        f.write(f"# all other kernels (torch input rendering etc.): TotalDurationNs={other_ns:.0f}\n")
    for r in keep:
        print(short(r["Name"]), "calls", r["Calls"], "avg_ns", r["AverageNs"], "min", r["MinNs"], "max", r["MaxNs"])


def pmc(d, out, skip):
    path = find(d, "*counter_collection.csv")
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = short(r.get("Kernel_Name", ""))
        if k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {}
    for k, cs in acc.items():
        res[k] = {}
        for c, vals in cs.items():
            v = vals[skip:] if len(vals) > skip else vals
            res[k][c] = dict(mean=sum(v) / len(v), min=min(v), max=max(v), dispatches=len(v))
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(res, sort_keys=True))


if __name__ == "__main__":
    mode, d, out = sys.argv[1:4]
    if mode == "stats":
        stats(d, out)
    else:
        pmc(d, out, int(sys.argv[4]) if len(sys.argv) > 4 else 0)
