"""The C++ host facade (include/SDF_Hashtable.h, libsdf_hashtable.so) in a plain C++ program,
the way the reference's Application.cpp uses its SDF_Hashtable: no Python, no torch."""
import os
import re
import subprocess

import numpy as np
import pytest

from voxelhashing_demo_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_facade_program(oracle, vh, torch_cuda, tmp_path):
    lib = os.path.join(ROOT, "voxelhashing_demo_amd", "lib")
    exe = tmp_path / "facade_demo"
    subprocess.run(["/opt/rocm/bin/hipcc", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "facade_demo.cpp"), "-o", str(exe),
                    "-L", lib, "-lsdf_hashtable", "-lvoxelhash_hip", f"-Wl,-rpath,{lib}"], check=True)
    verts = synth.sphere_inside_scene()
    path = tmp_path / "verts.bin"
    verts.tofile(path)
    out = subprocess.run([str(exe), str(path)], check=True, capture_output=True, text=True).stdout
    got = dict((k, int(v)) for k, v in re.findall(r"(\w+)=(\d+)", out))
    ot = oracle.OracleTable(oracle.default_params(), 640, 480, oracle.SEM_REFERENCE)   # common.h defaults
    I4 = np.eye(4, dtype=np.float32)
    ot.integrate(I4, verts)
    ot.integrate(I4, verts)
    assert got["allocated"] == len(ot.allocated()) > 50
    assert got["occupied"] == len(ot.compact())
    assert got["hits"] == int((ot.raycast(I4) > 0).sum())
