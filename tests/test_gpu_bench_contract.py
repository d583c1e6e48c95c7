"""bench.py's output contract, end to end on the GPU box: the one-GPU line and the N > 1 line the driver
launches through torch.distributed.run.  A one-GPU box cannot run two RCCL ranks (RCCL refuses two ranks on
one device), so the two-rank run uses bench.py's test rig: gloo as the transport (device buffers staged
through the host) and both ranks on cuda:0 -- the same sharded code path, windows, reductions and JSON."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def _line(out):
    """The driver's reading of stdout: the LAST non-empty line is the record, and it is short enough to survive the 8 KB tail
    the driver keeps (a 20 KB line left BENCH_r05 unparsed)."""
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert lines, out
    last = lines[-1]
    assert last.startswith("{") and len(last) < 4096, (len(last), last[:300])
    assert sum(1 for ln in lines if ln.startswith("{") and '"metric"' in ln) == 1, out[-2000:]
    return json.loads(last)


def _detail():
    return json.load(open(os.path.join(ROOT, "bench_detail.json")))


def _check(rec, n_gpus, steps, warmup):
    for k in REQUIRED:
        assert k in rec, k
    assert rec["n_gpus"] == n_gpus and rec["steps"] == steps and rec["warmup"] == warmup
    assert rec["unit"] == "frames/s" and rec["value"] > 0 and rec["higher_is_better"] is True
    assert rec["scaling"] == "weak" and rec["vs_baseline"] is None and rec["dtype"] == "f32" and rec["data"] == "synthetic"
    assert "workload" in rec["config"] and "model" not in rec["config"]
    rf = rec["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and 0 < rf["frac"] < 1
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3


def test_one_gpu_line_default_legs_as_the_driver_runs_it(torch_cuda):
    """`python3 bench.py --gpus 1 --steps 20 --warmup 5`, every default leg: the last stdout line is the compact record
    (contract keys, roofline, cpu_baseline, per-leg numbers); the prose is in bench_detail.json."""
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "20", "--warmup", "5"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    rec = _line(p.stdout)
    _check(rec, 1, 20, 5)
    assert rec["roofline"]["kernel"] == "frame_pipelined_kernel" and rec["timed_s"] >= 0.3
    assert abs(rec["value"] - 1e3 * rec["config"]["frames_per_step"] / rec["ms_per_step"]) < 0.01 * rec["value"]
    cb = rec["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and cb["unit"] == "frames/s" and cb["sample"]
    assert rec["raycast_mpix_per_s"] > 0
    legs = rec["legs"]
    for k in ("C1", "C3", "walk_free", "loaded", "sharded_world1", "closed_loop"):
        assert k in legs, (k, sorted(legs))
    assert legs["C1"]["cpu_baseline"]["cores"] == 1 and legs["C1"]["cpu_baseline"]["value"] > 0
    assert legs["C3"]["cpu_baseline"]["value"] > 0 and legs["C3"]["cpu_baseline"]["one_thread_frames_per_s"] > 0
    assert 0 < legs["C3"]["frac"] < 1 and legs["C3"]["value"] > 0
    assert legs["walk_free"]["bound"] == "latency+issue" and legs["walk_free"]["value"] > rec["value"]
    assert "error" not in legs["sharded_world1"] and legs["sharded_world1"]["value"] > 0
    cl = legs["closed_loop"]
    assert cl["value"] > 0 and cl["integrate_us"] > 0 and cl["raycast_us"] > 0 and cl["align_us"] > 0
    det = _detail()
    # (the loop's drift over its 59 frames depends on the rounding of the fp32 ICP sums: 28 - 35 mm over the partitions of
    # profiles/r06_closed_loop_drift.txt, VH_ICP_BLOCKS = 128 ... 256; 0.74 m of travel)
    assert det["value"] == rec["value"] and det["roofline"]["residency"] and det["closed_loop"]["max_drift_mm"] < 50
    assert det["configs"]["C1"]["cpu_baseline"]["cases"]["sphere_inside_reference"]["occupied_blocks"] == 136   # SURVEY.md 8(a) H9 probe
    assert det["configs"]["C1"]["cpu_baseline"]["cases"]["sphere_outside_reference"]["occupied_blocks"] == 44


def test_sensor_fed_workload_runs_its_legs(torch_cuda):
    """C3 / C4table / C5table keep uint16 sensor images resident; the legs that hand vertex maps to
    vh_preprocess, ICP or the oracle must render their own instead of indexing the sensor tensor."""
    p = subprocess.run([sys.executable, "bench.py", "--workload", "C3", "--frames", "6", "--steps", "2", "--warmup", "1",
                        "--legs", "sensor,raycast,next,cpu", "--cpu-frames", "2", "--raycast-steps", "3"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    rec = _line(p.stdout)
    assert rec["config"]["workload"].startswith("C3") and rec["cpu_baseline"]["value"] > 0 and rec["raycast_mpix_per_s"] > 0
    det = _detail()
    assert det["next_rows"]["preprocess"]["us_per_frame"] > 0 and "sensor_depth_input" not in det


@pytest.mark.parametrize("ranks", [2, 4])
def test_multi_rank_line_over_the_test_rig(torch_cuda, ranks):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, VH_BENCH_BACKEND="gloo", VH_BENCH_SHARE_GPU="1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks),
                        "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", str(ranks), "--steps", "3",
                        "--warmup", "1", "--legs", "cpu", "--cpu-frames", "6"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    rec = _line(p.stdout)
    _check(rec, ranks, 3, 1)
    assert rec["config"]["frames_per_step"] == ranks * rec["config"]["frames_per_camera_per_exchange"]
    assert rec["config"]["key_bin_overflows"] == 0 and rec["config"]["occupied_blocks_all_ranks"] > 0
    assert abs(rec["value"] - 1e3 * rec["config"]["frames_per_step"] / rec["ms_per_step"]) < 0.01 * rec["value"]
    assert rec["cpu_baseline"] and rec["cpu_baseline"]["value"] > 0


def test_gpus_2_as_typed_on_a_one_gpu_box(torch_cuda):
    """`python bench.py --gpus 2` with no launcher around it: the parent starts the ranks as a child process; on a box with
    one GPU the ranks say so (and nothing touched the GPU twice); with the test rig (gloo, both ranks on cuda:0) the same
    command yields the two-rank line."""
    import torch
    base = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--legs", "cpu", "--cpu-frames", "6"]
    if torch.cuda.device_count() < 2:
        p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600, env=base)
        assert p.returncode != 0 and f"2 GPUs requested (--gpus 2), {torch.cuda.device_count()} visible" in p.stderr
        assert "launch with" not in p.stderr
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900,
                       env=dict(base, VH_BENCH_BACKEND="gloo", VH_BENCH_SHARE_GPU="1"))
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    rec = _line(p.stdout)
    _check(rec, 2, 3, 1)
    assert rec["exchange_ranks"]["ranks"] == 2 and rec["cpu_baseline"]["value"] > 0
    assert _detail()["n_gpus"] == 2


def test_sharded_line_names_its_transport(torch_cuda):
    """One rank: the line says that NOTHING carried the exchange (a lone rank applies its frames straight from the send buffers,
    ADVICE round 4) and which transport would; it carries the per-exchange phase times and the prediction written beforehand."""
    p = subprocess.run([sys.executable, "bench.py", "--sharded", "--steps", "3", "--warmup", "1", "--legs", "none"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    rec = _line(p.stdout)
    er = rec["exchange_ranks"]
    assert er["ranks"] == 1 and er["transport"].startswith("none (one rank") and "rccl" in er["transport"] and "not run" in er["self_check"]
    ph = rec["exchange_phases_us"]
    assert ph["exchanges"] == 9 and ph["generate"] > 0 and ph["apply"] > ph["generate"] * 0.5 and ph["collectives"] < 10.0
    assert rec["predicted"]["reference_walk"]["frames_per_s"] > 0 and "before" in _detail()["predicted"]["note"]


@pytest.mark.parametrize("ranks", [2, 4])
def test_gpus_n_on_rccl_itself_with_every_rank_a_host_of_its_own(rccl_rig, ranks):
    """`python bench.py --gpus N` as the driver types it, on the DEFAULT backend (nccl = RCCL) and the native host
    (vh_dist_step_batch): RCCL only refuses two ranks of one host on one device, so with VH_BENCH_SHARE_GPU=1 every rank
    names a host of its own (NCCL_HOSTID) and the N ranks of this one-GPU box meet over RCCL's socket transport.  The line
    must say that RCCL carried the exchange between N ranks; its rate means nothing (shared GPU, sockets)."""
    base = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "VH_BENCH_BACKEND")}
    p = subprocess.run([sys.executable, "bench.py", "--gpus", str(ranks), "--steps", "3", "--warmup", "1", "--legs", "none"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900, env=dict(base, VH_BENCH_SHARE_GPU="1"))
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-5000:])
    rec = _line(p.stdout)
    _check(rec, ranks, 3, 1)
    assert rec["exchange_ranks"] == {"transport": "rccl", "ranks": ranks, "shared_gpu": True, "self_check": "passed (vh_dist_self_check)"}
    assert rec["exchange_phases_us"]["exchanges"] == 9 and rec["exchange_phases_us"]["collectives"] > 0
    assert rec["predicted"]["reference_walk"]["frames_per_s"] > 0
    assert "vh_dist_step_batch" in rec["exchange_host"]
    assert rec["config"]["key_bin_overflows"] == 0 and rec["config"]["occupied_blocks_all_ranks"] > 0
