#!/bin/bash
# Kernel trace of the sharded world-1 leg (bench.py --sharded): per-kernel totals and the timeline of one exchange batch.
# On the GPU box:  bash tools/trace_sharded.sh [tag]   -> gpurun_out/<tag>/kernel_stats_C2sharded.txt, timeline_C2sharded.txt
TAG=${1:-sh}; OUT=gpurun_out/$TAG; mkdir -p $OUT
T=/tmp/trace_$TAG; rm -rf $T; mkdir -p $T
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $T -- python3 bench.py --sharded --legs none --workload C2 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/trace_sharded.log 2>&1
python3 tools/prof_summary.py stats $T $OUT/kernel_stats_C2sharded.txt > /dev/null 2>&1
python3 - "$T" "$OUT/timeline_C2sharded.txt" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70], r.get("Stream_Id", "")))
rows.sort()
# one exchange batch from the middle of the run: from one key-generation launch to the next
gen = [i for i, r in enumerate(rows) if "generate_keys" in r[2]]
out = open(sys.argv[2], "w")
import statistics
fr = [r for r in rows if "frame_multi_pipelined" in r[2]]
if len(gen) < len(fr) // 64:
    # fused generation (vh_dist option "fused_generation", the default): in steady state there is no generation launch -- an
    # exchange is 8 frame launches on one stream.  The separate launches left are the first two exchanges behind each flush.
    print(f"fused generation: {len(gen)} key-generation launches beside {len(fr)} frame launches (the first two exchanges behind each flush)", file=out)
    m = len(fr) // 2
    t0 = fr[m][0]
    print("sixteen consecutive frame launches from the middle of the run (two exchanges):", file=out)
    for s, e, n, st in fr[m:m + 16]:
        print(f"  {(s - t0) / 1e3:8.1f} .. {(e - t0) / 1e3:8.1f} us  ({(e - s) / 1e3:6.1f})  stream {st:>3}  {n}", file=out)
    gaps = [(fr[i + 1][0] - fr[i][1]) / 1e3 for i in range(len(fr) - 1)]
    small = [g for g in gaps if g <= 2.0]
    big = [g for g in gaps if g > 2.0]
    dur = [(r[1] - r[0]) / 1e3 for r in fr]
    print(f"frame launches: {len(fr)}, mean {statistics.mean(dur):.2f} median {statistics.median(dur):.2f} us; gaps between consecutive ones inside a timed window (consecutive launches abut in the "
          f"trace: a launch's start stamp is its predecessor's end, so the durations carry the hand-over): mean {statistics.mean(small):.2f} us ({len(small)}); {len(big)} gaps > 2 us (window boundaries, flushes) totalling {sum(big):.0f} us", file=out)
    per = statistics.mean(dur) + statistics.mean(small)
    print(f"period per frame launch under the trace: {per:.2f} us = {1e6 / per:.0f} frames/s", file=out)
    out.close()
    print(open(sys.argv[2]).read())
    sys.exit(0)
if len(gen) > 12:
    a, b = gen[len(gen) // 2], gen[len(gen) // 2 + 1]
    t0 = rows[a][0]
    print(f"one exchange batch: {(rows[b][0] - t0) / 1e3:.1f} us from key generation to key generation", file=out)
    for s, e, n, st in rows[a:b]:
        print(f"  {(s - t0) / 1e3:8.1f} .. {(e - t0) / 1e3:8.1f} us  ({(e - s) / 1e3:6.1f})  stream {st:>3}  {n}", file=out)
# where the time between key generations goes, over the whole run: frame launches, gaps between them on their stream
iv = [(rows[gen[i + 1]][0] - rows[gen[i]][0]) / 1e3 for i in range(len(gen) - 1)]
gaps = [(fr[i + 1][0] - fr[i][1]) / 1e3 for i in range(len(fr) - 1)]
big = [g for g in gaps if g > 2.0]
print(f"batches: {len(iv)}; interval median {statistics.median(iv):.1f} mean {statistics.mean(iv):.1f} p90 {sorted(iv)[int(0.9 * len(iv))]:.1f} max {max(iv):.1f} us", file=out)
print(f"frame launches: {len(fr)}, mean {statistics.mean((r[1] - r[0]) / 1e3 for r in fr):.2f} us; gaps between consecutive ones: mean {statistics.mean(gaps):.2f} us, "
      f"{len(big)} gaps > 2 us totalling {sum(big):.0f} us = {sum(big) / max(1, len(iv)):.1f} us per batch", file=out)
out.close()
print(open(sys.argv[2]).read())
PY
