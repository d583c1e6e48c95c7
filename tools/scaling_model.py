#!/usr/bin/env python3
"""profiles/r05_scaling_model.json: predicted whole-job frames/s of `bench.py --gpus N` at N = 1, 2, 4, 8, written BEFORE any
multi-GPU measurement (the pool has one GPU per box), from
  * per-rank kernel times measured one rank at a time on one MI355X (tools/scaling_inputs.py),
  * the bytes a rank sends to each peer per exchange / a stated xGMI rate,
  * the world-1 efficiency of the native exchange measured by bench.py (sharded_world1 against its own kernel time).
Model of one exchange (B frames per camera; three streams overlap):
  apply   = B * (frame_us + gap_us)                       frame launches on the owner, back to back
  comm    = (bin_bytes + packet_bytes) / link_rate        every peer pair has a link of its own (8 GPUs fully connected: 7 links
                                                          per GPU), RCCL's all-to-all / all-gather drive them in parallel
  gen     = a shard of more than 60 MB (vh_dist option "fused_generation" 1, the default's size rule): rides in the frame launches,
            no term of its own, 3 % on apply; a smaller shard: launches of its own on a second stream, gen_us, 8 % on apply
  period  = max(apply * (1 + contention), comm[, gen])    contention: what the generation costs the frame launches
  frames/s = N * B / period
   tools/scaling_model.py gpurun_out/r05/scaling_inputs_C2.json [gpurun_out/r05/scaling_inputs_C5.json] > profiles/r05_scaling_model.json"""
import json
import sys

LINK_GBS = dict(conservative=45.0, nominal=64.0)      # achieved GB/s per direction of ONE xGMI link (peak ~76.8 of 153.6 bidirectional)
GAP_US = 1.0            # between two frame launches of a batch (profiles/r04_timeline_C2sharded.txt: 8.1 us per batch of 8)
CONTENTION_SEPARATE = 0.08
CONTENTION = 0.03       # the generating role aboard a frame launch: 19.2-19.3 us against 18.9 without it (profiles/r05_fused_generation_ab.txt);
                        # (0.08 for the generation as launches of its own on a second stream, the model's first version: 47.7 k at N = 1, measured 45.6 k)
out = dict(
    note="written before any multi-GPU run; inputs measured one rank at a time on one MI355X (tools/scaling_inputs.py)",
    assumptions=dict(link_gbs_per_direction=LINK_GBS, gap_us_per_launch=GAP_US, contention=dict(fused=CONTENTION, separate=CONTENTION_SEPARATE),
                     topology="every pair of the node's GPUs has one xGMI link; the collectives of an exchange use them in parallel, "
                              "so the wire time is (bytes to ONE peer) / (one link's rate) whatever N is",
                     scaling="weak: one camera per GPU, the logical table fixed: a rank's shard is 1/N of it"),
    workloads={})
for path in sys.argv[1:]:
    inp = json.load(open(path))
    B = inp["batch"]
    out["batch"] = B
    pred = {}
    for N, r in sorted(inp["ranks"].items(), key=lambda kv: int(kv[0])):
        n = int(N)
        e = {}
        for walk, key in (("reference_walk", "frame_us"), ("walk_free", "frame_index_us")):
            fused = r["shard_mb"] > 60.0 * 1.048576 and walk == "reference_walk"     # (vh_api_shard.hip: multi_fusing_pays; the walk-free launch never fuses)
            apply_us = B * (r[key] + GAP_US) * (1.0 + (CONTENTION if fused else CONTENTION_SEPARATE))
            row = dict(apply_us=round(apply_us, 1), generation="in the frame launches" if fused else "launches of its own", gen_us_if_launched_separately=r["gen_us"])
            for name, gbs in LINK_GBS.items():
                comm_us = 0.0 if n == 1 else (r["bin_bytes_per_peer"] + r["packet_bytes_per_peer"]) / (gbs * 1e3)
                period = max(apply_us, comm_us) if fused else max(apply_us, comm_us, r["gen_us"])
                row[name] = dict(comm_us=round(comm_us, 1), period_us=round(period, 1), frames_per_s=round(n * B / period * 1e6),
                                 bound="apply" if period == apply_us else "collectives" if period == comm_us else "generation")
            e[walk] = row
        e["inputs"] = r
        pred[N] = e
    base = pred["1"]["reference_walk"]["nominal"]["frames_per_s"]
    for N, e in pred.items():
        e["speedup_vs_1_nominal"] = round(e["reference_walk"]["nominal"]["frames_per_s"] / base, 2)
    out["workloads"][inp["workload"]] = dict(image=f'{inp["width"]}x{inp["height"]}', buckets=inp["buckets"], predicted=pred)
print(json.dumps(out, indent=1))
