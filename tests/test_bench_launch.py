"""`python bench.py --gpus N` as typed (no launcher around it): the parent starts the N ranks as a child process
(torch.distributed.run, one rank per GPU) without touching the GPU itself, relays the child's output and exit code.  Here,
without a GPU, every rank must say how many GPUs were asked for and how many it sees -- not print launcher instructions."""
import os
import subprocess

import pytest
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_launcher_command_is_the_drivers():
    import bench
    cmd = bench.launcher_command(["--gpus", "4", "--steps", "7", "--warmup", "2", "--workload", "C5"], 4, 29517)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29517"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "7", "--warmup", "2", "--workload", "C5"]     # arguments pass through unchanged
    # port 0 (what bench.py itself uses): the launcher picks the rendezvous port -- no port found free here and taken by another run since
    own = bench.launcher_command(["--gpus", "2"], 2, 0)
    assert "--master-port" not in own and "--rdzv-endpoint=127.0.0.1:0" in own and own[own.index("--local-addr") + 1] == "127.0.0.1"


@pytest.mark.parametrize("how", ["SIGTERM", "SIGKILL"])
def test_a_terminated_parent_takes_its_ranks_along(tmp_path, how):
    """SIGTERM to `python bench.py --gpus 2` (a driver's timeout): the launcher and its ranks end with it -- no orphan keeps a GPU
    busy.  SIGKILL (a harness's hard limit, the OOM killer) runs no handler in the parent: the launcher has asked the kernel for a
    SIGTERM at its parent's death (PR_SET_PDEATHSIG) and ends its ranks itself."""
    import signal
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HIP_VISIBLE_DEVICES"] = ""
    env["VH_BENCH_TEST_HOLD_S"] = "60"             # the ranks wait here before they look for a GPU (test hook in bench.py)
    p = subprocess.Popen([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1"], cwd=ROOT, env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    deadline = time.time() + 120
    def ranks():
        out = subprocess.run(["pgrep", "-f", "bench.py --gpus 2 --steps 3 --warmup 1"], capture_output=True, text=True).stdout.split()
        return [int(x) for x in out if int(x) != p.pid]
    while time.time() < deadline and len(ranks()) < 3:      # launcher + 2 ranks
        time.sleep(0.5)
    assert len(ranks()) >= 3, "the ranks did not start"
    p.send_signal(getattr(signal, how))
    assert p.wait(timeout=60) == (128 + signal.SIGTERM if how == "SIGTERM" else -signal.SIGKILL)
    end = time.time() + 40                           # (the ranks get SIGTERM from the parent's handler / from their launcher)
    while time.time() < end and ranks():
        time.sleep(0.5)
    assert ranks() == [], "ranks outlived their parent"


def test_gpus_2_without_a_launcher_starts_its_own_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HIP_VISIBLE_DEVICES"] = ""                # (whatever this machine has: the ranks must see fewer GPUs than asked)
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "torch.distributed.run" in p.stderr and "--nproc-per-node=2" in p.stderr      # the parent said what it started
    assert "2 GPUs requested (--gpus 2), 0 visible" in p.stderr
    assert "launch with" not in p.stderr and '"metric"' not in p.stdout


def test_world_size_mismatch_is_named():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "--gpus 2 but the launcher started 1 rank(s)" in p.stderr
