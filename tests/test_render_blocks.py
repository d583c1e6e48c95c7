"""Block silhouettes (SDFRenderer::drawToFrontAndBack, the reference's one working
render pass): per pixel the camera depth of the nearest front / farthest back face of the allocated
blocks' cubes.  Oracle behaviour on the CPU, bit-exact parity on the GPU."""
import numpy as np
import pytest

from voxelhashing_demo_amd import synth

W, H = 160, 120
KW = dict(numBuckets=1 << 12, numVoxelBlocks=4096)


def build(oracle, sem=1, n=4):
    ot = oracle.OracleTable(oracle.default_params(**KW), W, H, sem)
    prims = synth.room_primitives()
    poses = synth.camera_loop(40)
    frames = [(p, synth.render_room_verts(p, W, H, prims).numpy()) for p in poses[:n * 3:3]]
    for p, v in frames:
        ot.integrate(p if sem == 1 else np.eye(4, dtype=np.float32), v)
    return ot, frames


def test_silhouettes_bracket_the_raycast_surface(oracle):
    ot, frames = build(oracle)
    pose = frames[-1][0]
    front, back = ot.render_blocks(pose)
    ray = ot.raycast(pose)
    hit = ray > 0
    assert hit.sum() > 5000
    # wherever the raycast finds the surface it lies in an allocated block: between the two layers, up to
    # the half voxel by which the reference's cube [8k, 8k+8]*voxelSize is shifted against the voxel
    # centres 8k .. 8k+7 it stands for, plus the distance between a voxel's centre (where the DDA places
    # its sample) and the ray's passage through that voxel (two voxels of slack along a slanted ray)
    inside = hit & (front > 0) & (front <= ray + 0.04) & (back >= ray - 0.04)
    assert inside.sum() > 0.999 * hit.sum()              # (a few silhouette-edge pixels fall outside)
    assert ((front == 0) == (back == 0)).all() and (back >= front).all()
    # cubes are 16 cm: a single block seen head-on is at most sqrt(3)*0.16 deep along a ray
    solo = oracle.OracleTable(oracle.default_params(**KW), W, H, 1)
    v = np.zeros((H, W, 4), np.float32)
    v[H // 2, W // 2] = (0.0, 0.0, 1.0, 1.0)
    solo.integrate(np.eye(4, dtype=np.float32), v)
    f, b = solo.render_blocks(np.eye(4, dtype=np.float32))
    assert len(solo.allocated()) == 1 and 0 < (f > 0).sum() < 0.3 * W * H
    depth = (b - f)[f > 0]
    assert depth.max() <= np.sqrt(3) * 0.16 + 1e-4 and f[f > 0].min() >= 0.8


@pytest.mark.gpu
@pytest.mark.parametrize("sem", [0, 1])
def test_gpu_silhouettes_equal_oracle(oracle, vh, torch_cuda, sem):
    torch = torch_cuda
    ot, frames = build(oracle, sem)
    gt = vh.SDFHashtable(vh.default_params(**KW), W, H, sem)
    for p, v in frames:
        gt.integrate(p if sem == 1 else np.eye(4, dtype=np.float32), torch.from_numpy(v).cuda())
    front, back = torch.empty((H, W), device="cuda"), torch.empty((H, W), device="cuda")
    poses = [frames[-1][0], frames[0][0], np.eye(4, dtype=np.float32)]
    inside = np.asarray(frames[1][0], np.float32).reshape(4, 4).copy()
    if len(ot.allocated()):
        inside[:3, 3] = (ot.allocated()["pos"][0].astype(np.float32) * 8 + 4) * np.float32(0.02)   # a camera INSIDE a cube
        poses.append(inside)
    for pose in poses:
        for tmin, tmax in ((0.1, 5.0), (0.5, 2.0)):
            gt.render_blocks(pose, front, back, tmin, tmax)
            torch.cuda.synchronize()
            of, ob = ot.render_blocks(pose, tmin, tmax)
            assert np.array_equal(front.cpu().numpy().view(np.uint32), of.view(np.uint32))
            assert np.array_equal(back.cpu().numpy().view(np.uint32), ob.view(np.uint32))
    # the pass leaves the model alone
    gt.integrate(frames[0][0] if sem == 1 else np.eye(4, dtype=np.float32), torch.from_numpy(frames[0][1]).cuda())
    ot.integrate(frames[0][0] if sem == 1 else np.eye(4, dtype=np.float32), frames[0][1])
    gt.synchronize()
    assert np.array_equal(gt.hash_table()["pos"], ot.hash_table()["pos"])


@pytest.mark.gpu
def test_gpu_silhouettes_division_corner_cases(oracle, vh, torch_cuda):
    """The tile pass divides by a pixel's fixed ray direction through a hoisted reciprocal; everything the
    plain division would rescale takes the plain division.  Poses that reach those cases: an integer principal
    point with an axis-aligned camera (a column and a row of exactly-zero direction components), a camera
    centre exactly on cube faces (zero numerators) and 1e-30 / 1e-38 off them (tiny and denormal numerators),
    and a rotation by 1e-25 rad (direction components far below the fast range)."""
    torch = torch_cuda
    ot, frames = build(oracle, 1)
    gt = vh.SDFHashtable(vh.default_params(**KW), W, H, 1)
    for p, v in frames:
        gt.integrate(p, torch.from_numpy(v).cuda())
    for t in (ot, gt):
        t.set_raycast_intrinsics(140.0, 140.0, 80.0, 60.0)
    front, back = torch.empty((H, W), device="cuda"), torch.empty((H, W), device="cuda")
    corner = (ot.allocated()["pos"][len(ot.allocated()) // 2].astype(np.float32) * 8) * np.float32(0.02)
    poses = []
    for off in (0.0, 1e-30, -1e-38, 3e-20):
        p = np.eye(4, dtype=np.float32)
        p[:3, 3] = corner + np.float32(off)
        poses.append(p)
        q = p.copy()
        q[:3, 3] = (np.float32(off), corner[1], np.float32(1.0) + np.float32(off))
        poses.append(q)
    tilt = np.eye(4, dtype=np.float32)
    tilt[0, 2], tilt[2, 0] = np.float32(1e-25), np.float32(-1e-25)
    tilt[:3, 3] = (0.1, 1.4, 0.2)
    poses.append(tilt)
    yaw = np.asarray(frames[1][0], np.float32).reshape(4, 4).copy()
    poses.append(yaw)
    seen = 0
    for pose in poses:
        gt.render_blocks(pose, front, back, 0.0, 6.0)
        torch.cuda.synchronize()
        of, ob = ot.render_blocks(pose, 0.0, 6.0)
        assert np.array_equal(front.cpu().numpy().view(np.uint32), of.view(np.uint32))
        assert np.array_equal(back.cpu().numpy().view(np.uint32), ob.view(np.uint32))
        seen += int((ob > 0).sum())
    assert seen > 10 * W * H // 4


@pytest.mark.gpu
def test_gpu_silhouettes_of_cubes_that_straddle_the_near_plane(oracle, vh, torch_cuda):
    """A cube with corners nearer than t_min (or behind the camera) is bounded on the screen by its part beyond the plane
    z = 0.999 t_min: corners beyond it plus the points where edges cross it (vh_blocks.hip: block_bounds).  Cameras inside the
    model, at random attitudes, at and around allocated blocks, four near depths: every image equals the oracle's, which tests
    such a cube against every pixel of the image (vho_render_blocks: whole image as soon as a corner is nearer than 0.1)."""
    torch = torch_cuda
    ot, frames = build(oracle, 1)
    gt = vh.SDFHashtable(vh.default_params(**KW), W, H, 1)
    for p, v in frames:
        gt.integrate(p, torch.from_numpy(v).cuda())
    front, back = torch.empty((H, W), device="cuda"), torch.empty((H, W), device="cuda")
    blocks = ot.allocated()["pos"].astype(np.float64)
    rng = np.random.default_rng(11)
    straddling = 0
    for i in range(24):
        centre = (blocks[rng.integers(len(blocks))] * 8 + 4) * 0.02
        eye = centre + rng.uniform(-0.25, 0.25, 3) * (i % 3)          # inside a cube, then up to 0.25 / 0.5 m off it
        yaw, pitch, roll = rng.uniform(-np.pi, np.pi), rng.uniform(-1.2, 1.2), rng.uniform(-0.5, 0.5)
        cy, sy, cp, sp, cr, sr = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
        R = (np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]) @ np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
             @ np.array([[cr, -sr, 0], [sr, cr, 0], [0, 0, 1]]))
        pose = np.eye(4, dtype=np.float32)
        pose[:3, :3], pose[:3, 3] = R.astype(np.float32), eye.astype(np.float32)
        tmin = (0.05, 0.1, 0.3, 1.0)[i % 4]
        # (cubes with a corner nearer than t_min and one beyond it: the case under test)
        z = ((blocks[:, None, :] * 8 + 8 * np.array([[(c >> a) & 1 for a in range(3)] for c in range(8)])[None]) * 0.02 - eye) @ R[:, 2]
        straddling += int(((z.min(1) < 0.999 * tmin) & (z.max(1) > tmin)).sum())
        gt.render_blocks(pose, front, back, tmin, 6.0)
        torch.cuda.synchronize()
        of, ob = ot.render_blocks(pose, tmin, 6.0)
        assert np.array_equal(front.cpu().numpy().view(np.uint32), of.view(np.uint32)), (i, tmin)
        assert np.array_equal(back.cpu().numpy().view(np.uint32), ob.view(np.uint32)), (i, tmin)
    assert straddling > 100
    gt.close()
    ot.close()
