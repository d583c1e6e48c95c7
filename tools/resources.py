#!/usr/bin/env python3
"""Register / spill / occupancy summary of the kernels whose name contains one of the given substrings
(from voxelhashing_demo_amd/build/resource_usage.txt, written by `make -C voxelhashing_demo_amd/csrc asm`)."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
txt = open(os.path.join(ROOT, "voxelhashing_demo_amd", "build", "resource_usage.txt")).read()
want = sys.argv[1:] or [""]
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split()[0]
    if not any(w in name for w in want):
        continue
    def g(k):
        m = re.search(k + r": (\d+)", b)
        return int(m.group(1)) if m else None
    scratch, occ, lds = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")
    print(f"{name[:90]:90s} sgpr {g('SGPRs')} vgpr {g('VGPRs')} sspill {g('SGPRs Spill')} vspill {g('VGPRs Spill')} "
          f"scratch {scratch} occ {occ} lds {lds}")
