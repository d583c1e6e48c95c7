"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle on the same
inputs.  Bars (BASELINE.json north_star): allocated-block set and compact set
bit-exact (compared by block position -- which heap block a winner receives is
an atomic-order race in the reference too); TSDF values within
1e-4 (expected and also asserted: identical bits, contraction is off on both
sides).
"""
import numpy as np
import pytest

from conftest import blocks_by_pos, entries_as_set
from voxelhashing_demo_amd import synth

pytestmark = pytest.mark.gpu

TOL = 1e-4            # north_star: "TSDF within 1e-4 of reference"
I4 = np.eye(4, dtype=np.float32)


_FUSED = {"on": 1, "walk": 3}


@pytest.fixture(autouse=True, params=["fused-reference-walk", "fused-indexed-walk", "four-kernel-reference-walk"])
def frame_variant(request):
    """Every test runs in each variant of vh_integrate: the fused two-launch frame with the reference's walk over every
    VoxelEntry, the same with the walk over the bucket-occupancy bitmap (the library's default), and the four step kernels
    (alloc claim / commit / flatten / integrate) with the reference's walk (the steps with the bitmap walk: test_gpu_sequences.py)."""
    p = request.param
    _FUSED["on"] = 0 if p.startswith("four-kernel") else 1
    _FUSED["walk"] = 4 if "indexed" in p else 3
    yield p


def _pair(oracle, vh, sem, W=640, H=480, **over):
    kw = dict(numBuckets=1 << 17, numVoxelBlocks=4096)
    kw.update(over)
    ot = oracle.OracleTable(oracle.default_params(**kw), W, H, sem)
    gt = vh.SDFHashtable(vh.default_params(**kw), W, H, sem)
    gt.set_option("fused_frame", _FUSED["on"])
    gt.set_option("flatten_variant", _FUSED["walk"])
    return ot, gt


def _compare(ot, gt, exact=True):
    otab, gtab = ot.hash_table(), gt.hash_table()
    # same slots allocated, same keys in them (free slots carry the sentinel on both sides)
    assert np.array_equal(otab["ptr"] != -1, gtab["ptr"] != -1)
    assert np.array_equal(otab["pos"], gtab["pos"])
    assert np.array_equal(otab["offset"], gtab["offset"])
    galloc = gtab[gtab["ptr"] != -1]
    assert len(set(galloc["ptr"].tolist())) == len(galloc), "two entries share a voxel block"
    assert np.all(galloc["ptr"] % 512 == 0)
    ocomp, gcomp = ot.compact(), gt.compact()
    assert len(ocomp) == len(gcomp)
    assert entries_as_set(ocomp) == entries_as_set(gcomp)
    assert len(entries_as_set(gcomp)) == len(gcomp), "duplicate entry in the compact list"
    ovol, gvol = ot.sdf_blocks(), gt.sdf_blocks()
    ob = blocks_by_pos(otab[otab["ptr"] != -1], ovol)
    gb = blocks_by_pos(galloc, gvol)
    worst = 0.0
    for pos, ovox in ob.items():
        gvox = gb[pos]
        d = max(float(np.abs(ovox["sdf"] - gvox["sdf"]).max()), float(np.abs(ovox["weight"] - gvox["weight"]).max()))
        worst = max(worst, d)
        if exact:
            assert np.array_equal(ovox.view(np.uint32), gvox.view(np.uint32)), f"block {pos} differs in bits"
    assert worst <= TOL
    # voxels of blocks that were never handed out stay zero
    used = np.zeros(len(gvol) // 512, bool)
    used[galloc["ptr"] // 512] = True
    assert not gvol.view(np.uint32).reshape(-1, 1024)[~used].any()
    c = gt.counters()
    assert c["heap_counter"] == ot.heap_counter()
    assert c["allocated_total"] - c["freed_total"] == len(galloc)
    return worst


def _run(ot, gt, torch, frames):
    for pose, verts in frames:
        d_verts = torch.from_numpy(np.ascontiguousarray(verts)).cuda()
        ot.integrate(pose, verts)
        gt.integrate(pose, d_verts)
        gt.synchronize()


def test_sphere_inside_reference(oracle, vh, torch_cuda):
    """G2: camera inside a sphere, REFERENCE semantics, frames 0 and 1 (136 -> 151 blocks)."""
    ot, gt = _pair(oracle, vh, 0)
    verts = synth.sphere_inside_scene()
    _run(ot, gt, torch_cuda, [(I4, verts)])
    assert len(gt.allocated()) == 136
    _compare(ot, gt)
    _run(ot, gt, torch_cuda, [(I4, verts)])
    assert len(gt.allocated()) == 151 and gt.counters()["occupied"] == 151
    vol = gt.sdf_blocks()
    assert int((vol["weight"] > 0).sum()) == 72595
    _compare(ot, gt)


def test_sphere_outside_reference_touches_no_voxel(oracle, vh, torch_cuda):
    """G3: pins the K^T quirk -- 44 -> 47 blocks, zero voxels written."""
    ot, gt = _pair(oracle, vh, 0)
    verts = synth.sphere_outside_scene()
    _run(ot, gt, torch_cuda, [(I4, verts)])
    assert len(gt.allocated()) == 44
    _run(ot, gt, torch_cuda, [(I4, verts)])
    assert len(gt.allocated()) == 47
    assert not gt.sdf_blocks().view(np.uint32).any()
    _compare(ot, gt)


@pytest.mark.parametrize("sem", [0, 1])
def test_non_identity_pose(oracle, vh, torch_cuda, sem):
    """G4: 5 degree yaw + (0.1, 0, 0.05) m -- pins global_transform in alloc / frustum, the
    cofactor inverse and (REFERENCE) the voxel-index-space inverse in integrate."""
    ot, gt = _pair(oracle, vh, sem)
    pose = synth.yaw_pose(5.0, (0.1, 0.0, 0.05))
    verts = synth.sphere_inside_scene()
    _run(ot, gt, torch_cuda, [(pose, verts), (pose, verts), (I4, verts)])
    assert len(gt.allocated()) > 100
    _compare(ot, gt)


@pytest.mark.parametrize("sem", [0, 1])
def test_collision_stress_bucket_full(oracle, vh, torch_cuda, sem):
    """G5: 64 buckets x 2 slots: one insertion per bucket per frame, full buckets drop keys."""
    ot, gt = _pair(oracle, vh, sem, numBuckets=64, bucketSize=2, numVoxelBlocks=256)
    verts = synth.sphere_inside_scene()
    for frame in range(4):
        _run(ot, gt, torch_cuda, [(I4, verts)])
        n = len(gt.allocated())
        assert n <= 64 * min(frame + 1, 2)
        _compare(ot, gt)
    assert len(gt.allocated()) > 64          # second slots were used
    assert len(gt.allocated()) <= 128


def test_heap_exhaustion_is_bounded(oracle, vh, torch_cuda):
    """G5: the heap runs dry mid-frame.  The reference reads heap[-1] there; this build stops
    allocating.  Which of the frame's winners are refused is unspecified (atomic order), the
    count is exact and nothing is corrupted."""
    ot, gt = _pair(oracle, vh, 1, numVoxelBlocks=50)
    verts = synth.sphere_inside_scene()
    _run(ot, gt, torch_cuda, [(I4, verts), (I4, verts)])
    galloc, oalloc = gt.allocated(), ot.allocated()
    assert len(galloc) == 50 == len(oalloc)
    c = gt.counters()
    assert c["heap_counter"] == -1 and c["heap_exhausted"] > 0
    assert sorted(galloc["ptr"].tolist()) == [512 * i for i in range(50)]
    demanded = {oracle.world2block(p[:3], 0.02) for p in verts.reshape(-1, 4)}
    assert entries_as_set(galloc) <= demanded


def test_weight_saturation_ten_frames(oracle, vh, torch_cuda):
    """G6: running average and weight cap over 10 frames (weight = min(wmax, 0.1 n))."""
    ot, gt = _pair(oracle, vh, 1, integrationWeightMax=0.55)
    verts = synth.sphere_inside_scene()
    _run(ot, gt, torch_cuda, [(I4, verts)] * 10)
    _compare(ot, gt)
    w = gt.sdf_blocks()["weight"]
    assert float(w.max()) == pytest.approx(0.55)


def test_room_scene_moving_camera_pinhole(oracle, vh, torch_cuda):
    """C2-style input at reduced table size: box room, camera loop, PINHOLE semantics."""
    ot, gt = _pair(oracle, vh, 1, numVoxelBlocks=1 << 14)
    poses = synth.camera_loop(500)
    prims = synth.room_primitives()
    frames = [(poses[i], synth.render_room_verts(poses[i], prims=prims).numpy()) for i in (0, 1, 2, 40)]
    _run(ot, gt, torch_cuda, frames)
    assert len(gt.allocated()) > 300
    _compare(ot, gt)


def test_ragged_image_and_empty_frame(oracle, vh, torch_cuda):
    """Image size that is no multiple of the 16x16 tile nor of the wave; an all-invalid frame."""
    W, H = 200, 150
    ot, gt = _pair(oracle, vh, 1, W=W, H=H)
    empty = np.zeros((H, W, 4), np.float32)
    empty[..., 3] = 1.0
    _run(ot, gt, torch_cuda, [(I4, empty)])
    assert len(gt.allocated()) == 0 and gt.counters()["occupied"] == 0
    verts = synth.sphere_inside_scene(W, H)
    verts[::7, ::5, 2] = 0.0                      # holes
    _run(ot, gt, torch_cuda, [(I4, verts), (I4, verts)])
    assert len(gt.allocated()) > 20
    _compare(ot, gt)


def test_step_level_entry_points_match_fused(oracle, vh, torch_cuda):
    """reset / allocBlocks / flattenIntoBuffer / integrateDepthMap called one by one
    (VoxelUtils.h:5-13 order of SDF_Hashtable.cpp:11-40) equal the fused vh_integrate."""
    torch = torch_cuda
    ot, gt = _pair(oracle, vh, 0)
    verts = synth.sphere_inside_scene()
    d_verts = torch.from_numpy(verts).cuda()
    for _ in range(2):
        ot.integrate(I4, verts)
        gt.set_pose(I4)
        gt.reset_mutexes()
        gt.alloc_blocks(d_verts, None)
        n = gt.flatten(sync=True)
        assert n == len(ot.compact())
        gt.integrate_depth_map(d_verts)
        gt.synchronize()
    _compare(ot, gt)


def test_device_scalar_helpers(oracle, vh, torch_cuda):
    """Hash / rounding / projection / float->int on the device against the oracle, including
    negatives, half-way points, -0.0, NaN, infinities and values beyond the int range."""
    torch = torch_cuda
    rng = np.random.RandomState(7)
    pts = rng.uniform(-6, 6, size=(4096, 4)).astype(np.float32)
    vs = np.float32(0.02)
    special = np.array([
        [0.0, -0.0, 0.0, np.nan], [0.5 * vs, -0.5 * vs, 1.5 * vs, np.inf], [-0.31, 0.0, 0.009, -np.inf],
        [np.nan, 1.0, 1.0, 3e9], [np.inf, -np.inf, 1.0, -3e9], [1e12, -1e12, 5.0, 2147483520.0],
        [-0.16, -0.16, -0.16, -2147483648.0], [7.99 * vs, 8.0 * vs, 8.01 * vs, -2.9], [1e-30, -1e-30, 1e-38, 2.9],
    ], np.float32)
    pts = np.concatenate([special, pts])
    gt = vh.SDFHashtable(vh.default_params(numBuckets=1 << 17), 640, 480, 0)
    ot = oracle.OracleTable(oracle.default_params(numBuckets=1 << 17), 640, 480, 0)
    d_pts = torch.from_numpy(pts).cuda()
    d_out = torch.zeros((len(pts), 8), dtype=torch.int32, device="cuda")
    gt.debug_eval(d_pts, d_out)
    gt.synchronize()
    out = d_out.cpu().numpy()
    KT = synth.K_matrix(transposed=True)
    for i, p in enumerate(pts):
        b = oracle.world2block(p[:3], 0.02)
        assert tuple(out[i, :3]) == b, (i, p)
        assert out[i, 3] == oracle.hash_block(*b, 1 << 17)
        assert out[i, 4] == int(ot.block_in_frustum(b))
        assert tuple(out[i, 5:7]) == oracle.project(KT, p[:3]), (i, p)
        assert out[i, 7] == oracle.float2int_rz(p[3])


@pytest.mark.parametrize("mode", [1, 0])
def test_raycast_matches_oracle_and_geometry(oracle, vh, torch_cuda, mode):
    """Raycast spec (self-pinned, the reference's pass is disabled), voxel DDA (1) and fixed-step march (0):
    bit-equal to the oracle, and within a voxel of the analytic depth of the scene that was fused."""
    torch = torch_cuda
    ot, gt = _pair(oracle, vh, 1)
    ot.set_raycast_mode(mode)
    gt.set_raycast_mode(mode)
    verts = synth.sphere_inside_scene()
    _run(ot, gt, torch, [(I4, verts)] * 3)
    d_depth = torch.empty((480, 640), dtype=torch.float32, device="cuda")
    for pose in (I4, synth.yaw_pose(3.0, (0.02, 0.0, 0.01))):
        gt.raycast(pose, d_depth, 0.1, 5.0)
        gt.synchronize()
        g = d_depth.cpu().numpy()
        o = ot.raycast(pose, 0.1, 5.0)
        assert np.array_equal(g.view(np.uint32), o.view(np.uint32))
    g = d_depth.cpu().numpy()
    hit = g > 0
    assert hit.mean() > 0.5
    gt.raycast(I4, d_depth, 0.1, 5.0)
    gt.synchronize()
    g = d_depth.cpu().numpy()
    hit = g > 0
    z = verts[..., 2]
    assert np.abs(g[hit] - z[hit]).max() < 0.03      # 1.5 voxels (nearest-voxel sampling)


def test_drop_in_names(oracle, vh, torch_cuda):
    """The reference's own entry points (VoxelUtils.h:5-13) on the default context."""
    import ctypes as C
    torch = torch_cuda
    L = vh.load()
    p = vh.default_params(numBuckets=1 << 17, numVoxelBlocks=4096)
    L.updateConstantHashTableParams(C.byref(p))
    L.deviceAllocate(C.byref(p))
    L.calculateKinectProjectionMatrix()
    verts = synth.sphere_inside_scene()
    d_verts = torch.from_numpy(verts).cuda()
    ot = oracle.OracleTable(oracle.default_params(numBuckets=1 << 17, numVoxelBlocks=4096), 640, 480, 0)
    for _ in range(2):
        ot.integrate(I4, verts)
        L.updateConstantHashTableParams(C.byref(p))
        L.resetHashTableMutexes(C.byref(p))
        L.allocBlocks(d_verts.data_ptr(), None)
        n = L.flattenIntoBuffer(C.byref(p))
        p.numOccupiedBlocks = n
        L.updateConstantHashTableParams(C.byref(p))
        L.integrateDepthMap(C.byref(p), d_verts.data_ptr())
    assert n == 151
    gt = vh.SDFHashtable.__new__(vh.SDFHashtable)      # wrap the default context for the comparison
    gt._lib, gt._h, gt.params = L, C.c_void_p(L.vh_default_context()), p
    gt.bucket_range = (0, p.numBuckets)
    try:
        _compare(ot, gt)
    finally:
        gt._h = None
        L.deviceFree()


@pytest.mark.parametrize("mode", [1, 0])
def test_raycast_room_scene_moving_camera(oracle, vh, torch_cuda, mode):
    """Raycast of the fused room (thousands of blocks, rays crossing long stretches of empty
    space, so the empty-block / empty-macro-cell skips and the bucket bitmap are exercised) from poses on
    and off the integration path: bit-equal to the oracle (DDA: which leaves absent blocks only, through
    exact look-ups; fixed step: which evaluates every sample)."""
    torch = torch_cuda
    ot, gt = _pair(oracle, vh, 1, numVoxelBlocks=1 << 14)
    ot.set_raycast_mode(mode)
    gt.set_raycast_mode(mode)
    poses = synth.camera_loop(500)
    prims = synth.room_primitives()
    frames = [(poses[i], synth.render_room_verts(poses[i], prims=prims).numpy()) for i in (0, 3, 6, 30)]
    _run(ot, gt, torch, frames)
    d_depth = torch.empty((480, 640), dtype=torch.float32, device="cuda")
    for pose in (poses[3], poses[15], synth.yaw_pose(200.0, (0.3, 0.1, -0.4))):
        gt.raycast(pose, d_depth, 0.1, 5.0)
        gt.synchronize()
        g = d_depth.cpu().numpy()
        o = ot.raycast(pose, 0.1, 5.0)
        assert np.array_equal(g.view(np.uint32), o.view(np.uint32))
    gt.raycast(poses[3], d_depth, 0.1, 5.0)
    gt.synchronize()
    assert (d_depth.cpu().numpy() > 0).mean() > 0.3


@pytest.mark.parametrize("sem,band", [(0, 0.1), (1, 0.1), (1, 0.33)])
def test_truncation_band_allocation(oracle, vh, torch_cuda, sem, band):
    """Opt-in extension: blocks along the viewing ray within +-band of the
    surface are demanded too.  Same winner rule (launch order, then sample index); bit-equal."""
    ot, gt = _pair(oracle, vh, sem, numVoxelBlocks=1 << 14)
    ot.set_alloc_band(band)
    gt.set_alloc_band(band)
    pose = synth.yaw_pose(5.0, (0.1, 0.0, 0.05))
    verts = synth.sphere_inside_scene()
    poses = synth.camera_loop(500)
    prims = synth.room_primitives()
    frames = [(pose, verts), (pose, verts), (I4, verts)]
    if sem == 1:
        frames += [(poses[i], synth.render_room_verts(poses[i], prims=prims).numpy()) for i in (0, 1, 2)]
    _run(ot, gt, torch_cuda, frames)
    _compare(ot, gt)
    surface_only = oracle.OracleTable(oracle.default_params(numBuckets=1 << 17, numVoxelBlocks=1 << 14), 640, 480, sem)
    for p, v in frames:
        surface_only.integrate(p, v)
    assert entries_as_set(surface_only.allocated()) <= entries_as_set(gt.allocated()) or sem == 0
    assert len(gt.allocated()) > 1.3 * len(surface_only.allocated())
    with pytest.raises(vh.VoxelHashError):
        gt.set_alloc_band(10.0)          # more than 31 half-block steps


def test_negative_depth_vertices_are_processed_like_the_reference(oracle, vh, torch_cuda):
    """allocBlocksKernel only skips z == 0 (VoxelUtils.cu:621): vertices with a negative z are
    allocated for too (they just never pass the depth <= 0 test of the TSDF update)."""
    ot, gt = _pair(oracle, vh, 0)
    verts = synth.sphere_inside_scene().copy()
    verts[100:200, 100:300, :3] *= -1.0
    _run(ot, gt, torch_cuda, [(I4, verts), (I4, verts)])
    _compare(ot, gt)
    assert ot.last_stats["pixels_valid"] == 640 * 480


@pytest.mark.parametrize("sem", [0, 1])
def test_hostile_vertex_values(oracle, vh, torch_cuda, sem):
    """NaN, +-inf, denormals, huge magnitudes (block indices beyond the int range), negative and tiny
    depths and w != 1 sprinkled over a frame, with a pose that is not rigid: the GPU follows the oracle
    through every saturating conversion and wrapping product (SURVEY.md H2, H5, T2 sentinel)."""
    ot, gt = _pair(oracle, vh, sem, numBuckets=1 << 12, numVoxelBlocks=8192)
    verts = synth.sphere_inside_scene()
    rng = np.random.RandomState(11)
    H, W = verts.shape[:2]
    specials = np.array([np.nan, np.inf, -np.inf, 1e-42, -1e-42, 3e38, -3e38, 1e9, -1e9, 4.3e7, -4.3e7, 1e-7, -0.0,
                         2147483.6, -2147483.6], np.float32)
    for comp in range(4):
        ys, xs = rng.randint(0, H, 600), rng.randint(0, W, 600)
        verts[ys, xs, comp] = specials[rng.randint(0, len(specials), 600)]
    pose = np.array([[1.1, 0.05, 0, 0.1], [0, 0.9, 0.1, -0.05], [0.02, 0, 1.0, 0.2], [0, 0, 0, 1]], np.float32)
    _run(ot, gt, torch_cuda, [(I4, verts), (pose, verts), (I4, verts)])
    assert gt.counters()["heap_exhausted"] == 0
    _compare(ot, gt)


@pytest.mark.parametrize("sem", [0, 1])
def test_integrate_depth_equals_preprocess_plus_integrate(oracle, vh, torch_cuda, sem):
    """vh_integrate_depth: the frame straight from the uint16 sensor image (vertices computed inside the
    claim phase, TSDF update reading the image) against oracle preProcess + integrate."""
    torch = torch_cuda
    W, H = 320, 240
    ot, gt = _pair(oracle, vh, sem, W=W, H=H, numBuckets=1 << 14, numVoxelBlocks=8192)
    K = synth.K_matrix(W, H)
    kinv = np.linalg.inv(K.astype(np.float64)).astype(np.float32)
    prims = synth.room_primitives()
    poses = synth.camera_loop(60)
    for i in (0, 3, 6, 9, 3):
        z = synth.render_room_verts(poses[i], W, H, prims).numpy()[..., 2]
        d16 = np.round(z * 5000.0).clip(0, 65535).astype(np.uint16)
        d16[::11, ::5] = 0                                              # sensor holes
        verts = oracle.preprocess(d16, kinv)[0]
        ot.integrate(poses[i] if sem == 1 else I4, verts)
        gt.integrate_depth(poses[i] if sem == 1 else I4, torch.from_numpy(d16).cuda(), kinv)
    gt.synchronize()
    if sem == 1:
        assert len(gt.allocated()) > 300
    _compare(ot, gt)
