#!/usr/bin/env python3
"""profiles/pmc_latest.json from the condensed rocprofv3 --pmc passes (tools/profile_round.sh).

HBM bytes per launch = 2 * FETCH_SIZE[KB] * 1024 + WRITE_SIZE[KB] * 1024.  The factor 2 is the
gfx950 correction of MI355X_MICROARCH.md section HBM (FETCH_SIZE tallies 128-byte requests at
64 bytes).  Calibration in this access pattern: reset_table_kernel stores exactly
20*N = 104 857 600 bytes and WRITE_SIZE reports 104.89 MB; the table walk consumes every
fetched line, and 2*FETCH_SIZE matches its algorithmic bytes to < 1 %.

  make_pmc_latest.py gpurun_out/r01 C2 [more workloads...]
"""
import json
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
out_path = os.path.join(root, "profiles", "pmc_latest.json")
out = json.load(open(out_path)) if os.path.exists(out_path) else {}
for wl in sys.argv[2:]:
    f = json.load(open(os.path.join(src, f"pmc_FETCH_SIZE_{wl}.json")))
    w = json.load(open(os.path.join(src, f"pmc_WRITE_SIZE_{wl}.json")))
    d = {"source": f"{src}: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, bench.py --workload {wl} "
                   "--steps 100 --warmup 10, mean over dispatches after the first 10",
         "correction": "hbm_bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950: FETCH_SIZE counts 128-B requests as 64 B)"}
    for k in sorted(set(f) | set(w)):
        fk = f.get(k, {}).get("FETCH_SIZE", {}).get("mean", 0.0)
        wk = w.get(k, {}).get("WRITE_SIZE", {}).get("mean", 0.0)
        d[k + "_fetch_size_kb"] = round(fk, 2)
        d[k + "_write_size_kb"] = round(wk, 2)
        d[k + "_hbm_bytes_per_launch"] = int(round(2 * fk * 1024 + wk * 1024))
    out[wl] = d
json.dump(out, open(out_path, "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
