"""bench.py's stdout contract without a GPU: the compact line built from a full record (the 20 KB record of round 5, which the
driver could not parse as one line) carries the contract's keys, the roofline and the CPU baseline as numbers, and fits the
driver's tail with room to spare; emit() puts it LAST on stdout and the full record on stderr / in bench_detail.json."""
import io
import json
import os
import sys
from contextlib import redirect_stderr, redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def _record():
    return json.load(open(os.path.join(ROOT, "profiles", "r05_bench_default.json")))


def test_compact_line_of_a_full_record():
    full = _record()
    assert len(json.dumps(full)) > 16000
    line = bench.compact_line(full)
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) < bench.COMPACT_LIMIT - 512
    for k in CONTRACT:
        assert k in line, k
    assert line["value"] == full["value"] and line["config"]["workload"].startswith("C2")
    rf = line["roofline"]
    assert rf["bound"] == "hbm" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and "traffic" in rf
    assert all(not isinstance(v, str) or len(v) < 64 for v in rf.values())
    assert line["cpu_baseline"]["kind"] == "port" and len(line["cpu_baseline"]["sample"]) <= 120
    assert line["legs"]["C3"]["frac"] == full["configs"]["C3"]["roofline"]["frac"]
    assert line["legs"]["walk_free"]["bound"] == "latency+issue"
    assert line["legs"]["sharded_world1"]["value"] == full["sharded_world1"]["value"]


def test_compact_sharded_line():
    rec = _record()["sharded_world1"]
    line = bench.compact_sharded_line(rec)
    assert len(json.dumps(line, separators=(",", ":"))) < bench.COMPACT_LIMIT - 1024
    for k in CONTRACT:
        assert k in line, k
    assert line["config"]["frames_per_step"] == 8 and line["exchange_ranks"]["ranks"] == 1
    assert line["exchange_phases_us"]["exchanges"] == 9 and line["predicted"]["reference_walk"]["frames_per_s"] > 0


def test_emit_prints_the_line_last_and_trims_if_it_must(tmp_path, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    full = _record()
    line = bench.compact_line(full)
    line["legs"]["padding"] = "x" * 5000                     # a line that would not fit is cut down, loudly, never printed long
    out, err = io.StringIO(), io.StringIO()
    with redirect_stdout(out), redirect_stderr(err):
        bench.emit(full, line)
    lines = [ln for ln in out.getvalue().splitlines() if ln.strip()]
    assert len(lines) == 1 and len(lines[0]) < bench.COMPACT_LIMIT
    rec = json.loads(lines[0])
    assert rec["value"] == full["value"] and rec["roofline"] and rec["cpu_baseline"]
    assert "trimmed" in err.getvalue() and "bench.py detail: {" in err.getvalue()
    assert json.load(open(tmp_path / "bench_detail.json"))["value"] == full["value"]
