"""Pins the CPU oracle to the values the survey recorded from a host emulation of
the UNMODIFIED reference source (SURVEY.md section 8(a) [probe] values and
BASELINE.md section 1).  The reference itself has no tests or golden vectors
 and cannot be built here, so these anchors are the
strongest pin available; every number below is quoted from those two files.
"""
import numpy as np
import pytest

from voxelhashing_demo_amd import synth

NB = 1 << 17
I4 = np.eye(4, dtype=np.float32)


# ---- scalar KATs ----
def test_hash_probe_values(oracle):
    assert oracle.hash_block(2, 0, 8, NB) == 5378          # H1 [probe]
    assert oracle.hash_block(-6, -5, 10, NB) == 1075       # H1 [probe]


def test_hash_is_unsigned_modulo(oracle):
    # the xor is negative as an int for this key; the reference's modulo is unsigned
    x, y, z = -6, -5, 10
    h = ((x * 73856093) ^ (y * 19349669) ^ (z * 83492791)) & 0xFFFFFFFF
    assert oracle.hash_block(x, y, z, NB) == h % NB
    assert oracle.hash_block(x, y, z, 5000) == h % 5000


def test_world2voxel_probe_value(oracle):
    assert oracle.world2voxel((-0.31, 0.0, 0.009), 0.02) == (-16, 0, 0)   # H2 [probe]


def test_world2voxel_rounds_half_away_and_negative_zero(oracle):
    vs = np.float32(0.02)
    assert oracle.world2voxel((0.5 * vs, -0.5 * vs, 1.5 * vs), float(vs)) == (1, -1, 2)
    # copysignf(1, -0.0) = -1: -0.0 + -0.5 truncates to 0
    assert oracle.world2voxel((-0.0, 0.0, 0.0), 0.02) == (0, 0, 0)


def test_voxel2block_probe_value(oracle):
    assert oracle.voxel2block((-1, -8, -9)) == (-1, -1, -2)               # H3 [probe]
    assert oracle.voxel2block((0, 7, 8)) == (0, 0, 1)


def test_float2int_matches_gpu_cvt(oracle):
    assert oracle.float2int_rz(float("nan")) == 0
    assert oracle.float2int_rz(float("inf")) == 0x7FFFFFFF                # T2: (int)+inf on the GPU
    assert oracle.float2int_rz(float("-inf")) == -0x80000000
    assert oracle.float2int_rz(3e9) == 0x7FFFFFFF
    assert oracle.float2int_rz(-2.9) == -2
    assert oracle.float2int_rz(2.9) == 2


def test_projection_is_K_transposed_in_reference_mode(oracle):
    # H5: q = (fx*x, fy*y, cx*x + cy*y + z)
    KT = synth.K_matrix(transposed=True)
    p = np.array([0.3, -0.2, 1.7], np.float32)
    fx, fy, cx, cy = synth.intrinsics()
    qz = cx * p[0] + cy * p[1] + p[2]
    want = (int(np.trunc(fx * p[0] / qz)), int(np.trunc(fy * p[1] / qz)))
    assert oracle.project(KT, p) == want
    # 0/0 lands on pixel (0,0) and is accepted
    assert oracle.project(KT, (0.0, 0.0, 0.0)) == (0, 0)


def test_struct_layout(oracle):
    import ctypes
    assert oracle.ENTRY_DTYPE.itemsize == 20 and oracle.VOXEL_DTYPE.itemsize == 8   # fact 5
    assert ctypes.sizeof(oracle.Params) == 176


# ---- scene anchors ----
def _demanded_keys(oracle, verts):
    v = verts.reshape(-1, 4)
    v = v[v[:, 2] != 0]
    keys = {oracle.world2block(p[:3], 0.02) for p in v}
    return keys


def _run_two_frames(oracle, verts, sem):
    p = oracle.default_params(numBuckets=NB, numVoxelBlocks=4096)
    t = oracle.OracleTable(p, 640, 480, sem)
    counts = []
    for _ in range(2):
        t.integrate(I4, verts)
        counts.append(len(t.allocated()))
    vol = t.sdf_blocks()
    touched = vol["weight"] > 0
    rng = (float(vol["sdf"][touched].min()), float(vol["sdf"][touched].max())) if touched.any() else None
    passing = sum(t.block_in_frustum(k) for k in _demanded_keys(oracle, verts))
    out = dict(counts=counts, touched=int(touched.sum()), range=rng, passing=passing,
               occupied=len(t.compact()))
    t.close()
    return out


def test_sphere_outside_reference(oracle):
    verts = synth.sphere_outside_scene()
    assert len(_demanded_keys(oracle, verts)) == 71                       # "keys demanded"
    r = _run_two_frames(oracle, verts, oracle.SEM_REFERENCE)
    assert r["passing"] == 47
    assert r["counts"] == [44, 47]                                        # allocated after frame 0 -> 1
    assert r["touched"] == 0                                              # pins the K^T quirk


def test_sphere_inside_reference(oracle):
    verts = synth.sphere_inside_scene()
    assert int((verts[..., 2] != 0).sum()) == 307200                      # all pixels valid
    assert len(_demanded_keys(oracle, verts)) == 219
    r = _run_two_frames(oracle, verts, oracle.SEM_REFERENCE)
    assert r["passing"] == 151
    assert r["counts"] == [136, 151]
    assert r["touched"] == 72595
    assert r["range"][0] == pytest.approx(-0.4904, abs=5e-5)
    assert r["range"][1] == pytest.approx(0.1307, abs=5e-5)


def test_sphere_scenes_pinhole(oracle):
    r = _run_two_frames(oracle, synth.sphere_outside_scene(), oracle.SEM_PINHOLE)
    assert r["counts"] == [64, 71] and r["touched"] == 24663
    r = _run_two_frames(oracle, synth.sphere_inside_scene(), oracle.SEM_PINHOLE)
    assert r["counts"] == [157, 179] and r["touched"] == 88018
    assert r["range"][0] == pytest.approx(-0.1928, abs=5e-5)
    assert r["range"][1] == pytest.approx(0.1905, abs=5e-5)


def test_band_allocation_extends_the_surface_allocation(oracle):
    """alloc_band = 0 is the reference behaviour (all anchors above run with it); a positive band
    only adds blocks along the viewing rays."""
    verts = synth.sphere_inside_scene()
    tabs = {}
    for band in (0.0, 0.2):
        t = oracle.OracleTable(oracle.default_params(numBuckets=NB, numVoxelBlocks=1 << 14), 640, 480, oracle.SEM_PINHOLE)
        t.set_alloc_band(band)
        for _ in range(6):
            t.integrate(I4, verts)
        tabs[band] = {tuple(e["pos"]) for e in t.allocated()}
        t.close()
    assert len(tabs[0.0]) == 179                       # BASELINE.md anchor, unchanged
    assert tabs[0.0] < tabs[0.2] and len(tabs[0.2]) > 2 * len(tabs[0.0])
    # every extra block lies within the band of the R = 2 m sphere (block diagonal of slack)
    for pos in tabs[0.2]:
        centre = (np.array(pos, np.float64) * 8 + 3.5) * 0.02
        assert abs(np.linalg.norm(centre) - 2.0) < 0.2 + 0.3


def test_threaded_frame_is_identical(oracle):
    """vho_integrate_mt (bench.py's multi-thread CPU baseline) leaves the bits of vho_integrate."""
    import numpy as np
    from voxelhashing_demo_amd import synth
    W, H = 160, 120
    prims = synth.room_primitives()
    frames = [(p, synth.render_room_verts(p, W, H, prims).numpy()) for p in synth.camera_loop(30)[::5]]
    for sem, band in ((0, 0.0), (1, 0.0), (1, 0.1)):
        kw = dict(numBuckets=257, bucketSize=3, numVoxelBlocks=2048)
        a = oracle.OracleTable(oracle.default_params(**kw), W, H, sem)
        b = oracle.OracleTable(oracle.default_params(**kw), W, H, sem)
        a.set_alloc_band(band)
        b.set_alloc_band(band)
        for i, (p, v) in enumerate(frames):
            assert a.integrate(p, v) == b.integrate_mt(p, v, 1 + i % 5)
            assert a.last_stats == b.last_stats
        assert np.array_equal(a.hash_table(), b.hash_table()) and np.array_equal(a.compact(), b.compact())
        assert np.array_equal(a.sdf_blocks().view(np.uint32), b.sdf_blocks().view(np.uint32))
