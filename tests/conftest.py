import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def vh():
    """The product package, with the HIP library built if it is missing."""
    import voxelhashing_demo_amd as V
    from voxelhashing_demo_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    V.load()
    return V


@pytest.fixture(scope="session")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible (these tests never fall back to the CPU)")
    return torch


@pytest.fixture(scope="session")
def rccl_rig(torch_cuda):
    """The rig of tests/test_gpu_dist_rccl.py -- R rank processes on ONE GPU, each a host of its own to RCCL (NCCL_HOSTID), meeting
    over RCCL's socket transport on the loop-back interface -- checked once with torch's own collectives
    (tools/micro/rccl_one_gpu_probe.py).  Where the rig itself cannot run (no loop-back sockets for RCCL, an RCCL that compares
    devices differently) the tests that need it are skipped with that reason; where it runs, a failure in them is the library's."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    try:
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "micro", "rccl_one_gpu_probe.py")], env=env,
                           capture_output=True, text=True, timeout=400)
        ok, why = p.returncode == 0 and "all_reduce -> 3.0" in p.stdout, (p.stdout + p.stderr)[-800:]
    except subprocess.TimeoutExpired:
        ok, why = False, "the probe timed out"
    if not ok:
        pytest.skip("RCCL cannot run two ranks on this box's one GPU even with a host id per rank: " + why)
    return True


@pytest.fixture(scope="session", autouse=True)
def _room_frames_rendered_once():
    """The synthetic room frames of the tests (synth.render_room_verts) cost ~1 s each at 640x480 on the host, and the same few
    dozen (pose, size) pairs are rendered by test after test: on a GPU box they are ray-cast on the GPU (both sides of a
    parity test are handed the same array, so where it was computed is immaterial) and every frame is kept for the session.
    Without a GPU (the `-m "not gpu"` run) only the memo applies."""
    from collections import OrderedDict

    import torch

    from voxelhashing_demo_amd import synth
    plain = synth.render_room_verts
    on_gpu = torch.cuda.is_available()
    memo, limit = OrderedDict(), 1 << 30            # bytes of frames kept

    def render(pose, width=640, height=480, prims=None, device="cpu"):
        if str(device) != "cpu":
            return plain(pose, width, height, prims, device)
        prims = synth.room_primitives() if prims is None else prims
        key = (np.asarray(pose, np.float32).tobytes(), int(width), int(height),
               tuple((k, np.asarray(c, np.float64).tobytes(), np.asarray(sz, np.float64).tobytes()) for k, c, sz in prims))
        hit = memo.get(key)
        if hit is None:
            hit = plain(pose, width, height, prims, "cuda").cpu() if on_gpu else plain(pose, width, height, prims, "cpu")
            memo[key] = hit
            while sum(t.numel() * 4 for t in memo.values()) > limit and len(memo) > 1:
                memo.popitem(last=False)
        else:
            memo.move_to_end(key)
        return hit.clone()

    synth.render_room_verts = render
    yield
    synth.render_room_verts = plain


def entries_as_set(entries):
    """{(x,y,z)} of a VoxelEntry array."""
    return set(map(tuple, np.asarray(entries["pos"]).reshape(-1, 3).tolist()))


def blocks_by_pos(entries, volume):
    """{(x,y,z): 512 voxels} for a table / compact list and its volume."""
    out = {}
    for e in entries:
        p = int(e["ptr"])
        out[tuple(int(c) for c in e["pos"])] = volume[p:p + 512]
    return out
