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
    # the second table: integrateBatch + a pipelined frame + flush, then renderBlocks
    ot.integrate(I4, verts)
    assert got["allocated2"] == len(ot.allocated())
    assert got["covered"] == int((ot.render_blocks(I4)[0] > 0).sum()) > 1000
    # the third table: the walk-free frame through the facade (three frames = the oracle's table after its third)
    assert got["allocated3"] == len(ot.allocated()) and got["occupied3"] == len(ot.compact())


def test_cpp_tracking_program(oracle, vh, torch_cuda, tmp_path):
    lib = os.path.join(ROOT, "voxelhashing_demo_amd", "lib")
    exe = tmp_path / "tracking_demo"
    subprocess.run(["/opt/rocm/bin/hipcc", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "tracking_demo.cpp"), "-o", str(exe),
                    "-L", lib, "-lsdf_hashtable", "-lvoxelhash_hip", f"-Wl,-rpath,{lib}"], check=True)
    prims = synth.room_primitives()
    poses = synth.camera_loop(250)
    K = synth.K_matrix(640, 480)
    kinv = np.linalg.inv(K.astype(np.float64)).astype(np.float32)
    z0 = synth.render_room_verts(poses[100], 640, 480, prims).numpy()[..., 2]
    z1 = synth.render_room_verts(poses[101], 640, 480, prims).numpy()[..., 2]
    p0, n0 = oracle.depth_to_maps(z0, kinv)
    p1, _ = oracle.depth_to_maps(z1, kinv)
    for name, a in (("input", p1), ("target", p0), ("normals", n0)):
        a.tofile(tmp_path / f"{name}.bin")
    out = subprocess.run([str(exe)] + [str(tmp_path / f"{n}.bin") for n in ("input", "target", "normals")],
                         check=True, capture_output=True, text=True).stdout
    got = np.array([float(x) for x in out.splitlines()[0].split()]).reshape(4, 4)
    want, it, err, cnt = oracle.icp_align(p1, p0, n0, K, 0.08, 20, 0)
    true = np.linalg.inv(np.asarray(poses[100], np.float64).reshape(4, 4)) @ np.asarray(poses[101], np.float64).reshape(4, 4)
    assert np.abs(got - want).max() < 2e-4
    assert np.abs(got[:3, 3] - true[:3, 3]).max() < 5e-4
