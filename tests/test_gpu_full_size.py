"""Parity at BASELINE.json's full sizes.  C2 (2^20 buckets x 5, 2^18 blocks, 640x480) is small
enough for the oracle to follow for a few frames, so the HIP path is compared with it exactly;
C3 (2^22 buckets, 1280x960, 5 mm voxels) is compared exactly for two frames and then through
size-independent properties: idempotence of the allocated set under a repeated frame, heap
accounting, no duplicate key, weight = min(wmax, 0.1 n), compact count = allocated and in frustum."""
import numpy as np
import pytest

from conftest import entries_as_set
from test_gpu_parity import _compare
from voxelhashing_demo_amd import synth

pytestmark = pytest.mark.gpu


def _frames(W, H, idx, device=None):
    poses = synth.camera_loop(500)
    prims = synth.room_primitives()
    return [(poses[i], synth.render_room_verts(poses[i], W, H, prims).numpy()) for i in idx]


def test_c2_full_size_against_oracle(oracle, vh, torch_cuda):
    torch = torch_cuda
    kw = dict(numBuckets=1 << 20, numVoxelBlocks=1 << 18)
    ot = oracle.OracleTable(oracle.default_params(**kw), 640, 480, 1)
    gt = vh.SDFHashtable(vh.default_params(**kw), 640, 480, 1)
    for pose, verts in _frames(640, 480, (0, 1, 2, 3, 50, 51)):
        ot.integrate(pose, verts)
        gt.integrate(pose, torch.from_numpy(verts).cuda())
    gt.synchronize()
    _compare(ot, gt)
    assert len(gt.allocated()) > 500


@pytest.mark.parametrize("walk", [3, 4])
def test_c3_full_size_properties(oracle, vh, torch_cuda, walk):
    """walk 3: the reference's walk, two-launch frames.  walk 4: the walk-free frame PIPELINED at C3's size (flatten_variant 4
    through vh_integrate_batch: the build bench.py's configs.C3.occupancy_index_variant times; 2^22 buckets: 128 index tiles,
    a table beyond the Infinity Cache: the non-temporal build), exact for the first frames, properties after."""
    torch = torch_cuda
    W, H = 1280, 960
    kw = dict(numBuckets=1 << 22, numVoxelBlocks=1 << 16, voxelSize=0.005)
    ot = oracle.OracleTable(oracle.default_params(**kw), W, H, 1)
    gt = vh.SDFHashtable(vh.default_params(**kw), W, H, 1)
    gt.set_option("flatten_variant", walk)
    frames = _frames(W, H, (0, 1) if walk == 3 else (0, 1, 2, 3))
    if walk == 4:
        gt.set_option("pipeline", 1)
        d_frames = [torch.from_numpy(v).cuda() for _, v in frames]         # (alive until the synchronisation below)
        gt.integrate_batch([p for p, _ in frames], d_frames)
        for pose, verts in frames:
            ot.integrate_mt(pose, verts, 16)
    else:
        for pose, verts in frames:
            ot.integrate(pose, verts)
            gt.integrate(pose, torch.from_numpy(verts).cuda())
    gt.synchronize()
    otab, gtab = ot.hash_table(), gt.hash_table()
    assert np.array_equal(otab["pos"], gtab["pos"])                       # exact, 21 M entries
    assert gt.counters()["occupied"] == len(ot.compact())
    if walk == 4:            # the voxels of every 5th block, bit for bit (the pipelined TSDF update read the index walk's list)
        ovol, live = ot.sdf_blocks(), np.nonzero(gtab["ptr"] != -1)[0]
        for i in live[::5]:
            g = gt.block_voxels(int(gtab["ptr"][i]))
            o = ovol[int(otab["ptr"][i]):int(otab["ptr"][i]) + 512]
            assert np.array_equal(g.view(np.uint32), o.view(np.uint32))
        assert entries_as_set(gt.compact()) == entries_as_set(ot.compact())
    # -- properties from here on (no oracle) --
    pose, verts = frames[1]
    d_verts = torch.from_numpy(verts).cuda()
    prev = -1
    for _ in range(12):                                                    # repeat one frame until converged
        gt.integrate(pose, d_verts)
        n = gt.counters()["allocated_total"]
        if n == prev:
            break
        prev = n
    assert n == prev, "the allocated set did not converge under a repeated frame"
    tab = gt.hash_table()
    alloc = tab[tab["ptr"] != -1]
    c = gt.counters()
    assert len(alloc) == c["allocated_total"] == (1 << 16) - 1 - c["heap_counter"]     # heap accounting
    assert len(entries_as_set(alloc)) == len(alloc)                                    # no duplicate key
    assert len(set(alloc["ptr"].tolist())) == len(alloc) and np.all(alloc["ptr"] % 512 == 0)
    comp = gt.compact()
    assert entries_as_set(comp) <= entries_as_set(alloc) and len(comp) == c["occupied"]
    # every compact entry's corner projects into the image (PINHOLE frustum rule), and the rest do not
    fx, fy, cx, cy = synth.intrinsics(W, H)
    Tinv = oracle.invert4x4(pose)

    def visible(pos):
        w = np.concatenate([(pos * 8).astype(np.float32) * np.float32(0.005), np.ones((len(pos), 1), np.float32)], 1)
        cam = w @ Tinv.T
        with np.errstate(all="ignore"):
            u = np.trunc((fx * cam[:, 0] + cx * cam[:, 2]) / cam[:, 2])
            v = np.trunc((fy * cam[:, 1] + cy * cam[:, 2]) / cam[:, 2])
        return (cam[:, 2] > 0) & (u >= 0) & (u < W) & (v >= 0) & (v < H)

    vis = visible(alloc["pos"])
    borderline = 50                  # float32 numpy vs the kernel's exact operation order at the image edge
    assert abs(int(vis.sum()) - len(comp)) <= borderline
    # weights: a voxel updated by k frames holds min(wmax, sum of k times 0.1f)
    vol = gt.sdf_blocks()
    w = vol["weight"]
    steps = np.cumsum(np.full(64, np.float32(0.1), np.float32), dtype=np.float32)
    assert np.isin(w[w > 0], steps).all()
    assert np.isfinite(vol["sdf"]).all() and float(np.abs(vol["sdf"]).max()) <= 1.0


def test_c2_snapshot_round_trip_and_resume(vh, torch_cuda, tmp_path):
    """C2-size model (2^20 buckets, 2^18 blocks): save -> load into a second context -> both continue with
    the same frames (one of them pipelined, with a collection in between): tables, free lists and every
    allocated block stay identical.  A round trip and a resume, GPU against GPU."""
    torch = torch_cuda
    kw = dict(numBuckets=1 << 20, numVoxelBlocks=1 << 18)
    a = vh.SDFHashtable(vh.default_params(**kw), 640, 480, 1)
    frames = [(p, torch.from_numpy(v).cuda()) for p, v in _frames(640, 480, tuple(range(0, 60, 3)))]
    for pose, dv in frames[:12]:
        a.integrate(pose, dv)
    path = tmp_path / "c2.snap"
    a.save_snapshot(path)
    b = vh.SDFHashtable(vh.default_params(**kw), 640, 480, 1)
    b.load_snapshot(path)
    ta, tb = a.hash_table(), b.hash_table()
    assert np.array_equal(ta, tb) and (ta["ptr"] != -1).sum() > 400          # the round trip: ptr included
    b.set_option("pipeline", 1)
    for t in (a, b):
        for pose, dv in frames[12:16]:
            t.integrate(pose, dv)
        t.garbage_collect(0.5)
        for pose, dv in frames[16:]:
            t.integrate(pose, dv)
        t.synchronize()
    ta, tb = a.hash_table(), b.hash_table()
    assert np.array_equal(ta["pos"], tb["pos"]) and np.array_equal(ta["ptr"] != -1, tb["ptr"] != -1)
    ca, cb = a.counters(), b.counters()
    for k in ("occupied", "allocated_total", "freed_total", "heap_counter", "heap_exhausted"):
        assert ca[k] == cb[k], k
    pa = {tuple(e["pos"]): int(e["ptr"]) for e in ta[ta["ptr"] != -1]}
    pb = {tuple(e["pos"]): int(e["ptr"]) for e in tb[tb["ptr"] != -1]}
    for key in list(pa)[::9]:
        va, vb = a.block_voxels(pa[key]), b.block_voxels(pb[key])
        assert np.array_equal(va["sdf"].view(np.uint32), vb["sdf"].view(np.uint32)) and np.array_equal(va["weight"], vb["weight"])
    a.close()
    b.close()
