"""Depth pre-processing: uint16 depth -> vertex + normal maps, the step
right upstream of integrate() (preProcess, CameraTrackingUtils.cu:50-120).  CPU: the oracle's
restatement against closed-form expectations.  GPU: the fused HIP kernel, bit-equal to the oracle,
and feeding vh_integrate exactly like a vertex map built on the host."""
import numpy as np
import pytest

from voxelhashing_demo_amd import synth

W, H = 640, 480


def k_inverse():
    fx, fy, cx, cy = synth.intrinsics(W, H)
    one = np.float32(1)
    return np.array([[one / fx, 0, -cx / fx], [0, one / fy, -cy / fy], [0, 0, 1]], np.float32)


def depth_u16(scene="inside"):
    z = synth.sphere_depth((0, 0, 0), 2.0, True) if scene == "inside" else synth.sphere_depth((0, 0, 1.5), 0.5, False)
    return np.round(z * 5000.0).astype(np.uint16)


def test_oracle_vertices_follow_the_reference_formula(oracle):
    d = depth_u16("outside")
    pos, nrm = oracle.preprocess(d, k_inverse())
    kinv = k_inverse()
    for (y, x) in [(240, 320), (100, 200), (0, 0), (479, 639), (250, 333)]:
        depth = np.float32(d[y, x]) / np.float32(5000.0)
        row = lambda r: kinv[r, 0] * np.float32(x) + kinv[r, 1] * np.float32(y) + kinv[r, 2] * np.float32(1)
        want = [row(0) * depth, row(1) * depth, row(2) * depth, np.float32(1)]
        assert pos[y, x].tolist() == [float(w) for w in want]
    invalid = d == 0
    assert invalid.any() and np.all(pos[invalid] == np.array([0, 0, 0, 1], np.float32))     # z == 0 <=> skipped by integrate


def test_oracle_normals(oracle):
    d = depth_u16("inside")
    pos, nrm = oracle.preprocess(d, k_inverse())
    assert not nrm[0].any() and not nrm[-1].any() and not nrm[:, 0].any() and not nrm[:, -1].any()   # border
    inner = nrm[1:-1, 1:-1, :3]
    length = np.linalg.norm(inner.astype(np.float64), axis=-1)
    assert np.all((np.abs(length - 1.0) < 1e-5) | (length == 0))
    assert (length > 0).mean() > 0.95
    assert not nrm[..., 3].any()
    # a sphere seen from its centre: the normal is (anti)parallel to the viewing ray
    v = pos[240, 320, :3] / np.linalg.norm(pos[240, 320, :3])
    assert abs(abs(float(np.dot(v, nrm[240, 320, :3]))) - 1.0) < 2e-2
    # a hole knocks out its own normal and its four neighbours'
    d2 = d.copy()
    d2[200, 300] = 0
    _, n2 = oracle.preprocess(d2, k_inverse())
    for (y, x) in [(200, 300), (199, 300), (201, 300), (200, 299), (200, 301)]:
        assert not n2[y, x].any()
    assert n2[198, 300].any()


@pytest.mark.gpu
@pytest.mark.parametrize("scene", ["inside", "outside"])
def test_hip_preprocess_bit_equal(oracle, vh, torch_cuda, scene):
    torch = torch_cuda
    from voxelhashing_demo_amd.hashtable import preprocess
    d = depth_u16(scene)
    d[::17, ::13] = 0
    opos, onrm = oracle.preprocess(d, k_inverse())
    d_depth = torch.from_numpy(d.astype(np.int16)).cuda()          # same 16 bits; torch has no CUDA uint16 ops
    gpos = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
    gnrm = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
    preprocess(d_depth, k_inverse(), gpos, gnrm)
    torch.cuda.synchronize()
    assert np.array_equal(gpos.cpu().numpy().view(np.uint32), opos.view(np.uint32))
    assert np.array_equal(gnrm.cpu().numpy().view(np.uint32), onrm.view(np.uint32))


@pytest.mark.gpu
def test_preprocess_feeds_integrate(oracle, vh, torch_cuda):
    """depth -> vh_preprocess -> vh_integrate on the GPU equals oracle.preprocess -> oracle.integrate,
    also through the reference's own names SetCameraIntrinsic / preProcess."""
    import ctypes as C
    torch = torch_cuda
    d = depth_u16("inside")
    kinv = k_inverse()
    opos, _ = oracle.preprocess(d, kinv)
    L = vh.load()
    kflat = np.ascontiguousarray(kinv.reshape(9))
    assert L.SetCameraIntrinsic(kflat.ctypes.data_as(C.POINTER(C.c_float)), kflat.ctypes.data_as(C.POINTER(C.c_float)))
    d_depth = torch.from_numpy(d.astype(np.int16)).cuda()
    gpos = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
    gnrm = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
    L.preProcess(gpos.data_ptr(), gnrm.data_ptr(), d_depth.data_ptr())
    kw = dict(numBuckets=1 << 17, numVoxelBlocks=4096)
    gt = vh.SDFHashtable(vh.default_params(**kw), W, H, 1)
    ot = oracle.OracleTable(oracle.default_params(**kw), W, H, 1)
    I4 = np.eye(4, dtype=np.float32)
    for _ in range(2):
        gt.integrate(I4, gpos, gnrm)
        ot.integrate(I4, opos)
    gt.synchronize()
    assert np.array_equal(gt.hash_table()["pos"], ot.hash_table()["pos"])
    assert len(gt.allocated()) > 100 and gt.counters()["occupied"] == len(ot.compact())
