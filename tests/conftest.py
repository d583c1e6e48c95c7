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
