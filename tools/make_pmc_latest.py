#!/usr/bin/env python3
"""profiles/pmc_latest.json from the condensed rocprofv3 --pmc passes (tools/profile_round.sh).

HBM bytes per launch = 2 * FETCH_SIZE[KB] * 1024 + WRITE_SIZE[KB] * 1024.  The factor 2 is the
gfx950 correction of MI355X_MICROARCH.md section HBM (FETCH_SIZE tallies 128-byte requests at
64 bytes).  Calibration in this access pattern: reset_table_kernel stores exactly
20*N = 104 857 600 bytes and WRITE_SIZE reports 104.89 MB; the table walk consumes every
fetched line, and 2*FETCH_SIZE matches its algorithmic bytes to < 1 %.

  make_pmc_latest.py gpurun_out/r02 C2 C3 [C2band C2sharded ...]

Extras picked up when present in the source directory:
  pmc_VALU_raycast_C2.json          -> "C2_raycast": VALU instructions per wave of raycast_kernel
  bench_sharded_world1_<WL>.json    -> "<WL>sharded".algorithmic_bytes_per_launch (the one-rank sharded run's
                                       roofline.bytes_per_launch, for the ratio bench.py applies at N > 1)
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
    d = {"source": f"{src}: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, bench.py --legs none --workload "
                   f"{wl} --steps 100 --warmup 10, mean over dispatches after the first 10",
         "correction": "hbm_bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950: FETCH_SIZE counts 128-B requests as 64 B)"}
    for k in sorted(set(f) | set(w)):
        fk = f.get(k, {}).get("FETCH_SIZE", {}).get("mean", 0.0)
        wk = w.get(k, {}).get("WRITE_SIZE", {}).get("mean", 0.0)
        d[k + "_fetch_size_kb"] = round(fk, 2)
        d[k + "_write_size_kb"] = round(wk, 2)
        d[k + "_hbm_bytes_per_launch"] = int(round(2 * fk * 1024 + wk * 1024))
    if wl.endswith("sharded"):
        try:
            line = json.load(open(os.path.join(src, f"bench_sharded_world1_{wl[:-7]}.json")))
            d["algorithmic_bytes_per_launch"] = line["roofline"]["bytes_per_launch"]
            # key without template arguments, as dist.bench_sharded looks it up
            for k in list(d):
                for base in ("frame_multi_scan_claim_kernel", "frame_multi_pipelined_kernel"):
                    if k.startswith(base) and k.endswith("_hbm_bytes_per_launch"):
                        d[base + "_hbm_bytes_per_launch"] = d[k]
        except Exception as e:
            d["algorithmic_bytes_per_launch_error"] = repr(e)
    out[wl] = d
    valu = os.path.join(src, f"pmc_VALU_raycast_{wl}.json")
    if os.path.exists(valu):
        v = json.load(open(valu))
        # (the raycast leg also times the fixed-step march on the same poses: the DDA kernel, the default, is the one reported)
        for k, cs in sorted(v.items(), key=lambda kv: 0 if kv[0].startswith("raycast_coop_kernel<false>") else 1 if kv[0].startswith(("raycast_coop", "raycast_dda")) else 2):
            if k.startswith(("raycast_coop_kernel", "raycast_dda_kernel", "raycast_kernel")) and "SQ_INSTS_VALU" in cs and "SQ_WAVES" in cs:
                out[wl + "_raycast"] = {
                    "source": f"{src}: rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES, bench.py --legs raycast --workload {wl}",
                    "kernel": k.split("(")[0],
                    "raycast_kernel_valu_insts": cs["SQ_INSTS_VALU"]["mean"], "raycast_kernel_waves": cs["SQ_WAVES"]["mean"],
                    "raycast_kernel_valu_per_wave": round(cs["SQ_INSTS_VALU"]["mean"] / max(1.0, cs["SQ_WAVES"]["mean"]), 1)}
                break
json.dump(out, open(out_path, "w"), indent=1, sort_keys=True)
# profiles/kernel_stats_latest.json: average dispatch duration per kernel and workload from the same round's
# rocprofv3 --kernel-trace --stats summaries (kernel_stats_<WL>.csv), for bench.py's frac_at_rocprofv3_mean
import csv
ks_path = os.path.join(root, "profiles", "kernel_stats_latest.json")
ks = json.load(open(ks_path)) if os.path.exists(ks_path) else {}
for wl in sys.argv[2:]:
    f = os.path.join(src, f"kernel_stats_{wl}.csv")
    if not os.path.exists(f):
        continue
    rows = [r for r in csv.DictReader(l for l in open(f) if not l.startswith("#"))]
    d = {"source": f"{f}: rocprofv3 --kernel-trace --stats of bench.py --legs none --workload {wl}"}
    for r in rows:
        name = r["Name"]
        i = name.find("vh::")
        short = name[i + 4:].split("(")[0] if i >= 0 else name
        d[short] = {"calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 3)}
    ks[wl] = d
json.dump(ks, open(ks_path, "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
