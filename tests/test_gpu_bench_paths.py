"""Every number of the bench line has an oracle test at its own configuration and call form.

bench.py times, on C2 (BASELINE.json configs[1]: 640x480, 2^20 buckets x 5, 2^18 voxel blocks, 2 cm voxels, PINHOLE),
  value / roofline        vh_integrate_batch(8) over consecutive frames of the 500-pose loop
  loaded_integrate        the same with vh_set_alloc_band(0.1), band_mode VH_BAND_RAY_DDA (and VH_BAND_RAY, its variant)
  sensor_depth_input      vh_integrate_depth_batch / vh_integrate_depth on uint16 sensor images
  sharded_world1          vh_dist_step_batch (key bins + sensor packets) with one rank over RCCL, and with two ranks over the
                          loop-back transport (the N > 1 lines)
  raycast                 vh_raycast on the model those frames built
Here each of them runs from a FRESH table (so the frames that insert blocks are inside) over 96 frames of that loop in
batches of 8, next to the oracle (vho_integrate_mt: identical results on several host threads), and is compared slot for
slot, every 7th block bit for bit, the compact set, the counters, and one raycast (depth and normals)."""
import ctypes as C

import numpy as np
import pytest

from conftest import entries_as_set
from voxelhashing_demo_amd import synth

pytestmark = pytest.mark.gpu
W, H = 640, 480
KW = dict(numBuckets=1 << 20, numVoxelBlocks=1 << 18)
FRAMES, BATCH = 96, 8
THREADS = 16


@pytest.fixture(scope="module")
def loop_frames():
    """The first 96 poses of the C2 loop and their vertex maps, rendered on the host (the GPU and the oracle must consume
    the same bits: torch renders the scene with a last-ulp difference between CPU and GPU)."""
    poses = synth.camera_loop(500)[:FRAMES]
    prims = synth.room_primitives()
    return poses, [synth.render_room_verts(p, W, H, prims).numpy() for p in poses]


def _compare_light(ot, gt, every=7, min_blocks=500):
    """_compare of test_gpu_parity without the 1 GiB volume download: all slots, every `every`-th block's bits."""
    otab, gtab = ot.hash_table(), gt.hash_table()
    assert np.array_equal(otab["ptr"] != -1, gtab["ptr"] != -1)
    assert np.array_equal(otab["pos"], gtab["pos"])
    assert np.array_equal(otab["offset"], gtab["offset"])
    live = np.nonzero(gtab["ptr"] != -1)[0]
    assert len(live) >= min_blocks
    assert len(set(gtab["ptr"][live].tolist())) == len(live), "two entries share a voxel block"
    ovol = ot.sdf_blocks()
    for i in live[::every]:
        g = gt.block_voxels(int(gtab["ptr"][i]))
        o = ovol[int(otab["ptr"][i]):int(otab["ptr"][i]) + 512]
        assert np.array_equal(g.view(np.uint32), o.view(np.uint32)), f"block {tuple(gtab['pos'][i])} differs in bits"
    ocomp, gcomp = ot.compact(), gt.compact()
    assert len(ocomp) == len(gcomp) and entries_as_set(ocomp) == entries_as_set(gcomp)
    c = gt.counters()
    assert c["heap_counter"] == ot.heap_counter() and c["heap_exhausted"] == 0 and c["cand_overflow"] == 0
    assert c["allocated_total"] - c["freed_total"] == len(live)
    return len(live)


def _raycast_equal(ot, gt, torch, pose):
    d = torch.empty((H, W), dtype=torch.float32, device="cuda")
    n = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
    gt.raycast_normals(pose, d, n)
    gt.synchronize()
    od, on = ot.raycast(pose, normals=True)
    assert np.array_equal(d.cpu().numpy().view(np.uint32), od.view(np.uint32))
    assert np.array_equal(n.cpu().numpy().view(np.uint32), on.view(np.uint32))
    assert (od > 0).mean() > 0.5


@pytest.mark.parametrize("band,mode,walk", [(0.0, 0, 3), (0.1, 2, 3), (0.1, 0, 3), (0.0, 0, 4), (0.1, 2, 4)])
def test_c2_integrate_batch_from_a_fresh_table(oracle, vh, torch_cuda, loop_frames, band, mode, walk):
    """`value` (band 0), `loaded_integrate` (band 0.1 by the block DDA along the viewing ray, VH_BAND_RAY_DDA) and its
    `ray_samples_variant` (VH_BAND_RAY): vh_integrate_batch(8), pipelined frames, fresh table.  walk 4: the same as
    `occupancy_index_variant` runs them -- the walk-free frame (flatten_variant 4: its lean build without a band, the generic
    build with one)."""
    torch = torch_cuda
    poses, verts = loop_frames
    ot = oracle.OracleTable(oracle.default_params(**KW), W, H, 1)
    gt = vh.SDFHashtable(vh.default_params(**KW), W, H, 1)
    gt.set_option("pipeline", 1)                       # as bench.py's Integrator does
    gt.set_option("flatten_variant", walk)
    if band:
        ot.set_alloc_band(band, mode)
        gt.set_alloc_band(band)
        gt.set_option("band_mode", mode)
    d_verts = [torch.from_numpy(v).cuda() for v in verts]
    torch.cuda.synchronize()
    for k in range(0, FRAMES, BATCH):
        gt.integrate_batch(poses[k:k + BATCH], d_verts[k:k + BATCH])
        for j in range(k, k + BATCH):
            ot.integrate_mt(poses[j], verts[j], THREADS)
    gt.synchronize()
    blocks = _compare_light(ot, gt, min_blocks=(2000 if mode == 0 else 1500) if band else 500)
    assert gt.counters()["epoch"] == FRAMES
    _raycast_equal(ot, gt, torch, poses[40])
    print(f"band {band}: {blocks} blocks after {FRAMES} frames")
    gt.close()
    ot.close()


@pytest.mark.parametrize("batched", [True, False])
def test_c2_sensor_depth_frames_from_a_fresh_table(oracle, vh, torch_cuda, loop_frames, batched):
    """`sensor_depth_input`: the frames straight from uint16 sensor images (vh_integrate_depth_batch of 8 / vh_integrate_depth
    one by one, pipelined), against the oracle fed with vho_preprocess's vertex maps of the same images."""
    torch = torch_cuda
    poses, verts = loop_frames
    kinv = np.linalg.inv(synth.K_matrix(W, H).astype(np.float64)).astype(np.float32)
    d16 = [np.round(v[..., 2] * 5000.0).clip(0, 65535).astype(np.uint16) for v in verts[:48]]
    ot = oracle.OracleTable(oracle.default_params(**KW), W, H, 1)
    gt = vh.SDFHashtable(vh.default_params(**KW), W, H, 1)
    gt.set_option("pipeline", 1)
    dd = [torch.from_numpy(d).cuda() for d in d16]
    torch.cuda.synchronize()
    for k in range(0, 48, BATCH):
        if batched:
            gt.integrate_depth_batch(poses[k:k + BATCH], dd[k:k + BATCH], kinv)
        else:
            for j in range(k, k + BATCH):
                gt.integrate_depth(poses[j], dd[j], kinv)
        for j in range(k, k + BATCH):
            ot.integrate_mt(poses[j], oracle.preprocess(d16[j], kinv)[0], THREADS)
    gt.synchronize()
    _compare_light(ot, gt)
    _raycast_equal(ot, gt, torch, poses[20])
    gt.close()
    ot.close()


@pytest.mark.parametrize("world,walk", [(1, 3), (2, 3), (1, 4), (2, 4)])
def test_c2_sharded_native_exchange(oracle, vh, torch_cuda, loop_frames, world, walk):
    """`sharded_world1` and the N > 1 lines: the bucket-range-sharded path as bench.py runs it -- vh_dist_step_batch inside the
    library (key generation from the sensor images into ONE bin per (owner, batch), all-to-all of the bins, all-gather of the
    sensor packets, vh_apply_frames_batch with the batch's last frame deferred into the next exchange) at C2's table size,
    batches of 8, the library's default bin size.  world 1: over RCCL (ncclCommInitRank with one rank, as the bench leg);
    world 2: two ranks of this process over the loop-back transport, each camera feeding its own 48 frames of the loop.
    walk 4: the shards run the walk-free multi-camera frame (flatten_variant 4: the occupancy-index walk for all cameras), as
    `sharded_world1.occupancy_index_variant` does."""
    from test_gpu_configs import assert_slice_equals, compare_blocks
    from voxelhashing_demo_amd import dist as vdist
    torch = torch_cuda
    poses, verts = loop_frames
    kinv = np.linalg.inv(synth.K_matrix(W, H).astype(np.float64)).astype(np.float32)
    n = 48
    # camera r walks frames r*n .. r*n+n-1 of the loop (world 1: the bench's single camera)
    d16 = [[np.round(v[..., 2] * 5000.0).clip(0, 65535).astype(np.uint16) for v in verts[r * n:(r + 1) * n]] for r in range(world)]
    pre = [[oracle.preprocess(d, kinv)[0] for d in d16[r]] for r in range(world)]
    cam_poses = [poses[r * n:(r + 1) * n] for r in range(world)]
    dd = [[torch.from_numpy(d).cuda() for d in d16[r]] for r in range(world)]
    torch.cuda.synchronize()
    ot = oracle.OracleTable(oracle.default_params(**KW), W, H, 1)
    if world == 1:
        ranks = [vdist.NativeDist(vh.default_params(**KW), W, H, 1, 0, 1, BATCH, vdist.unique_id(), sensor_k_inv=kinv)]
        assert ranks[0].transport == "rccl" and ranks[0].comm_info() == (0, 1)
        group = None
    else:
        group = vdist.NativeGroup(vh.default_params(**KW), W, H, 1, world, BATCH, sensor_k_inv=kinv)
        ranks = group.ranks
    for nd in ranks:
        nd.table.set_option("flatten_variant", walk)
    for k in range(0, n, BATCH):
        if group:
            group.step([cam_poses[r][k:k + BATCH] for r in range(world)], [dd[r][k:k + BATCH] for r in range(world)])
        else:
            ranks[0].step(cam_poses[0][k:k + BATCH], dd[0][k:k + BATCH])
        for j in range(k, k + BATCH):
            if world == 1:
                ot.integrate_mt(cam_poses[0][j], pre[0][j], THREADS)       # one camera: the multi-camera frame is integrate()
            else:
                vdist.reference_multi_camera_frame(ot, [cam_poses[r][j] for r in range(world)], [pre[r][j] for r in range(world)])
    for nd in ranks:
        nd.flush()
    torch.cuda.synchronize()
    if world == 1:
        assert ranks[0].table.counters()["bin_overflow"] == 0
        _compare_light(ot, ranks[0].table)
    else:
        plan = vdist.ShardPlan(KW["numBuckets"], world)
        otab, total = ot.hash_table(), 0
        for r, nd in enumerate(ranks):
            lo, hi = plan.bucket_range(r)
            gtab = nd.table.hash_table()
            assert_slice_equals(gtab, otab, lo, hi, 5, f"shard {r}")
            total += compare_blocks(nd.table, gtab, ot, otab[lo * 5:hi * 5], every=7)
            c = nd.table.counters()
            assert c["bin_overflow"] == 0 and c["heap_exhausted"] == 0 and c["epoch"] == n
        assert total > 100
    # the raycast round through the shards
    outs = [torch.empty((H, W), dtype=torch.float32, device="cuda") for _ in range(world)]
    view_poses = [cam_poses[r][20] for r in range(world)]
    if group:
        group.raycast(view_poses, outs, 8192)
    else:
        ranks[0].raycast(view_poses[0], outs[0], 8192)
    for nd in ranks:
        nd.flush()
    torch.cuda.synchronize()
    for r in range(world):
        want = ot.raycast(view_poses[r])
        assert np.array_equal(outs[r].cpu().numpy().view(np.uint32), want.view(np.uint32)) and (want > 0).mean() > 0.5
    if group:
        group.close()
    else:
        ranks[0].close()
    ot.close()
