"""The opt-in extensions of SURVEY.md 8(f) next #2 on the GPU, against the oracle: overflow linked
list (slot-exact including the chain links in VoxelEntry::offset), normal-directed block DDA band,
depth-dependent truncation and sample weight.  tests/test_overflow_cpu.py pins the oracle's side."""
import numpy as np
import pytest

from test_gpu_parity import _compare
from test_overflow_cpu import _plane_scene, chains_ok
from voxelhashing_demo_amd import dist as vdist
from voxelhashing_demo_amd import synth

pytestmark = pytest.mark.gpu
I4 = np.eye(4, dtype=np.float32)
VARIANTS = {"fused-ballot-walk": (1, 3), "four-kernel-reference-walk": (0, 3), "fused-indexed-walk": (1, 4),
            "pipelined-ballot-walk": (1, 3, 1), "pipelined-indexed-walk": (1, 4, 1)}


def pair(oracle, vh, variant, W=640, H=480, sem=0, overflow=True, **kw):
    ot = oracle.OracleTable(oracle.default_params(**kw), W, H, sem)
    gt = vh.SDFHashtable(vh.default_params(**kw), W, H, sem)
    fused, walk = VARIANTS[variant][:2]
    gt.set_option("fused_frame", fused)
    gt.set_option("flatten_variant", walk)
    if len(VARIANTS[variant]) > 2:           # one launch per frame: with the overflow list, serialised inside the launch
        gt.set_option("pipeline", 1)
        gt.set_option("pipeline_overflow", 2)    # (... whatever the launch's size: by default only small launches take that form)
    if overflow:
        ot.set_overflow(True)
        gt.set_option("overflow_list", 1)
    gt.set_option("pipeline_overflow", 2)
    return ot, gt


@pytest.mark.parametrize("variant", list(VARIANTS))
@pytest.mark.parametrize("nb,bs,L", [(48, 5, 6), (64, 4, 4), (96, 2, 8)])
def test_overflow_list_collision_scene(oracle, vh, torch_cuda, variant, nb, bs, L):
    """G5's collision scene with the list on: frame by frame the same slots, the same chain links."""
    torch = torch_cuda
    ot, gt = pair(oracle, vh, variant, numBuckets=nb, bucketSize=bs, numVoxelBlocks=1024, attachedLinkedListSize=L)
    verts = synth.sphere_inside_scene()
    d_verts = torch.from_numpy(verts).cuda()
    prev = -1
    for f in range(40):
        ot.integrate(I4, verts)
        gt.integrate(I4, d_verts)
        gt.synchronize()
        _compare(ot, gt)
        n = len(gt.allocated())
        if n == prev:
            break
        prev = n
    assert n == prev and (gt.hash_table()["offset"] != 0).sum() >= 5
    chains_ok(oracle, ot)
    if (nb, bs, L) == (48, 5, 6):
        assert n == 150
    assert gt.counters()["cand_overflow"] == 0


@pytest.mark.parametrize("walk", [3, 4])
@pytest.mark.parametrize("nb,bs,L,chunk", [(48, 5, 6, 5), (96, 2, 8, 3), (64, 4, 4, 1)])
def test_overflow_list_pipelined_batches(oracle, vh, torch_cuda, walk, nb, bs, L, chunk):
    """One launch per frame with the list on: the claim and walk workgroups of frame i+1 wait inside the launch until
    commit(i) has published its tag, then read the settled table.  Batches without an observer in between (so that
    commit(i) and claim(i+1) really share launches), the collision scene and then a moving camera."""
    torch = torch_cuda
    kw = dict(numBuckets=nb, bucketSize=bs, numVoxelBlocks=1024, attachedLinkedListSize=L)
    ot = oracle.OracleTable(oracle.default_params(**kw), 640, 480, 1)
    gt = vh.SDFHashtable(vh.default_params(**kw), 640, 480, 1)
    ot.set_overflow(True)
    gt.set_option("overflow_list", 1)
    gt.set_option("pipeline_overflow", 2)                 # (1 200 claim tiles: beyond what the default takes as one launch)
    gt.set_option("flatten_variant", walk)
    gt.set_profiling(True)
    sphere = synth.sphere_inside_scene()
    from test_gpu_pipeline import room_frames
    frames = [(I4, sphere)] * 7 + room_frames(torch, 640, 480, (0, 1, 2, 3, 4, 8, 9))
    refused = 0
    for s0 in range(0, len(frames), chunk):
        part = frames[s0:s0 + chunk]
        gt.integrate_batch([p for p, _ in part], [torch.from_numpy(np.ascontiguousarray(v)).cuda() for _, v in part])
        for p, v in part:
            ot.integrate(p, v)
            refused += ot.last_stats["heap_exhausted"]
        _compare(ot, gt)
    chains_ok(oracle, ot)
    t = gt.kernel_times()
    assert t["frame_pipelined_ms"] > 0.0 and t["frame_commit_integrate_ms"] == 0.0
    assert (gt.hash_table()["offset"] != 0).sum() >= 3
    assert gt.counters()["heap_exhausted"] == refused == 0


@pytest.mark.parametrize("variant", ["fused-ballot-walk", "four-kernel-reference-walk", "pipelined-ballot-walk"])
def test_overflow_delete_and_collect(oracle, vh, torch_cuda, variant):
    """Deleting heads with followers, chained entries and plain slots; then garbage collection; then
    fusing on -- the table stays equal to the oracle's slot for slot."""
    torch = torch_cuda
    ot, gt = pair(oracle, vh, variant, numBuckets=48, bucketSize=5, numVoxelBlocks=1024, attachedLinkedListSize=6)
    verts = synth.sphere_inside_scene()
    d_verts = torch.from_numpy(verts).cuda()
    for _ in range(12):
        ot.integrate(I4, verts)
        gt.integrate(I4, d_verts)
    gt.synchronize()
    _compare(ot, gt)
    tab = ot.hash_table()
    heads = [i for i in range(4, len(tab), 5) if tab[i]["ptr"] != -1 and tab[i]["offset"] != 0]
    chained = [int(i) for i in np.nonzero(tab["ptr"] != -1)[0]
               if oracle.hash_block(*[int(c) for c in tab[i]["pos"]], 48) != i // 5]
    plain = [int(i) for i in np.nonzero(tab["ptr"] != -1)[0] if i % 5 == 2][:5]
    assert len(heads) >= 3 and len(chained) >= 5
    doomed = sorted(set(heads[:3] + chained[1::2] + plain))
    keys = np.zeros((len(doomed) + 1, 4), np.int32)
    keys[:-1, :3] = tab["pos"][doomed]
    keys[-1, :3] = (77, 77, 77)                                       # absent
    freed = ot.delete_blocks([tuple(k[:3]) for k in keys.tolist()])
    gt.delete_blocks(torch.from_numpy(keys).cuda())
    gt.synchronize()
    assert gt.counters()["last_freed"] == freed == len(doomed)
    _compare(ot, gt)
    chains_ok(oracle, ot)
    for _ in range(6):                                                # the keys come back
        ot.integrate(I4, verts)
        gt.integrate(I4, d_verts)
    gt.synchronize()
    _compare(ot, gt)
    # collection of everything the last frame saw that holds no surface (threshold 0.05)
    a = ot.garbage_collect(0.05)
    gt.garbage_collect(0.05)
    gt.synchronize()
    assert gt.counters()["last_freed"] == a > 0
    _compare(ot, gt)
    chains_ok(oracle, ot)
    for _ in range(4):
        ot.integrate(I4, verts)
        gt.integrate(I4, d_verts)
    gt.synchronize()
    _compare(ot, gt)
    # pool partition on the GPU side
    alloc = gt.allocated()
    heap = gt.heap()[:gt.counters()["heap_counter"] + 1]
    assert len(set(heap.tolist()) | set((alloc["ptr"] // 512).tolist())) == 1024 and len(heap) + len(alloc) == 1024


def test_overflow_raycast_and_snapshot(oracle, vh, torch_cuda, tmp_path):
    """Lookups follow the chains: the raycast of a table with chains equals the oracle's; a snapshot
    carries the chains and refuses to load into a context without the list."""
    torch = torch_cuda
    W, H = 320, 240
    kw = dict(numBuckets=512, bucketSize=2, numVoxelBlocks=4096, attachedLinkedListSize=8)
    ot, gt = pair(oracle, vh, "fused-ballot-walk", W, H, 1, **kw)
    poses = synth.camera_loop(60)
    prims = synth.room_primitives()
    for i in range(0, 24, 3):
        v = synth.render_room_verts(poses[i], W, H, prims).numpy()
        ot.integrate(poses[i], v)
        gt.integrate(poses[i], torch.from_numpy(v).cuda())
    gt.synchronize()
    _compare(ot, gt)
    assert (gt.hash_table()["offset"] != 0).sum() > 20
    depth = torch.empty((H, W), dtype=torch.float32, device="cuda")
    gt.raycast(poses[21], depth)
    gt.synchronize()
    ref = ot.raycast(poses[21])
    assert (ref > 0).mean() > 0.2 and np.array_equal(depth.cpu().numpy().view(np.uint32), ref.view(np.uint32))
    snap = tmp_path / "chains.vhsnap"
    gt.save_snapshot(snap)
    plain = vh.SDFHashtable(vh.default_params(**kw), W, H, 1)
    with pytest.raises(vh.VoxelHashError, match="overflow"):
        plain.load_snapshot(snap)
    plain.close()
    again = vh.SDFHashtable(vh.default_params(**kw), W, H, 1)
    again.set_option("overflow_list", 1)
    again.load_snapshot(snap)
    v = synth.render_room_verts(poses[27], W, H, prims).numpy()
    ot.integrate(poses[27], v)
    again.integrate(poses[27], torch.from_numpy(v).cuda())
    again.synchronize()
    _compare(ot, again)
    with pytest.raises(vh.VoxelHashError, match="before the first frame"):
        again.set_option("overflow_list", 0)


@pytest.mark.parametrize("world,calls,batch", [(2, "batched", 1), (4, "stepwise", 1), (2, "batched", 3)])
def test_overflow_on_shards(oracle, vh, torch_cuda, world, calls, batch):
    """Bucket-range shards with the list on: chains wrap inside a shard; R HIP shards equal ONE oracle
    table whose chains wrap inside segments of the shard size; the raycast over the shards (view
    tables with chains of their own) equals the oracle's raycast of that table."""
    torch = torch_cuda
    W, H = 160, 120
    kw = dict(numBuckets=512, bucketSize=2, numVoxelBlocks=4096, attachedLinkedListSize=8)
    plan = vdist.ShardPlan(kw["numBuckets"], world)
    # (batch 3: multi-camera frames share launches -- commit(b) with claim(b+1), serialised inside the launch by the list)
    shards = [vdist.HipShard(vh.default_params(**kw), W, H, 1, plan, r, W * H, batch=batch, batched_calls=(calls == "batched"))
              for r in range(world)]
    full = oracle.OracleTable(oracle.default_params(**kw), W, H, 1)
    full.set_overflow(True, plan.per_shard)
    for sh in shards:
        sh.table.set_option("overflow_list", 1)
        sh.table.set_option("pipeline_overflow", 2)
    prims = synth.room_primitives()
    poses = None
    for step in range(0, 6, batch):
        frames = []
        for b in range(batch):
            cams = []
            for r in range(world):
                pose = synth.camera_loop(40, phase=vdist.camera_phase(r, world))[(3 * (step + b)) % 40]
                cams.append((pose, synth.render_room_verts(pose, W, H, prims).numpy()))
            frames.append(cams)
        vdist.loopback_step(shards, [[frames[b][r][0] for b in range(batch)] for r in range(world)],
                            [[torch.from_numpy(frames[b][r][1]).cuda() for b in range(batch)] for r in range(world)])
        for cams in frames:
            vdist.reference_multi_camera_frame(full, [c[0] for c in cams], [c[1] for c in cams])
        poses = [c[0] for c in frames[-1]]
    ftab, fvol = full.hash_table(), full.sdf_blocks()
    total = 0
    for r, sh in enumerate(shards):
        sh.table.synchronize()
        lo, hi = plan.bucket_range(r)
        mine = sh.table.hash_table()
        want = ftab[lo * 2:hi * 2]
        assert np.array_equal(mine["pos"], want["pos"]) and np.array_equal(mine["offset"], want["offset"])
        assert np.array_equal(mine["ptr"] != -1, want["ptr"] != -1)
        for i in np.nonzero(mine["ptr"] != -1)[0][::3]:
            assert np.array_equal(sh.table.block_voxels(int(mine["ptr"][i])).view(np.uint32),
                                  fvol[int(want["ptr"][i]):int(want["ptr"][i]) + 512].view(np.uint32))
        total += int((mine["ptr"] != -1).sum())
        assert sh.table.counters()["bin_overflow"] == 0
    assert total == len(full.allocated()) and (ftab["offset"] != 0).sum() > 10
    views = [vdist.HipViewTable(vh.default_params(**kw), W, H, 1, world, 4096) for _ in range(world)]
    for v in views:
        v.table.set_option("overflow_list", 1)
    depths = vdist.loopback_raycast(shards, views, poses, capacity=4096)
    for r in range(world):
        assert views[r].table.counters()["bin_overflow"] == 0
        assert np.array_equal(depths[r].view(np.uint32), full.raycast(poses[r]).view(np.uint32)), r


@pytest.mark.parametrize("variant", list(VARIANTS))
def test_normal_dda_band(oracle, vh, torch_cuda, variant):
    """Band allocation by the block DDA along the normal: a tilted plane with an analytic normal map,
    then the room with the normal maps preProcess makes, moving camera -- slot-exact."""
    torch = torch_cuda
    W, H = 160, 120
    verts, normals, _ = _plane_scene(W, H)
    ot, gt = pair(oracle, vh, variant, W, H, 1, overflow=False, numBuckets=1 << 14, numVoxelBlocks=1 << 14)
    ot.set_alloc_band(0.2, oracle.BAND_NORMAL_DDA)
    gt.set_alloc_band(0.2)
    gt.set_option("band_mode", 1)
    dv, dn = torch.from_numpy(verts).cuda(), torch.from_numpy(normals).cuda()
    for _ in range(4):
        ot.integrate(I4, verts, normals)
        gt.integrate(I4, dv, dn)
        gt.synchronize()
        _compare(ot, gt)
    assert len(gt.allocated()) > 300
    # room scene: sensor depth -> preProcess maps on the GPU, the same bits handed to the oracle
    W, H = 320, 240
    ot, gt = pair(oracle, vh, variant, W, H, 1, overflow=False, numBuckets=1 << 15, numVoxelBlocks=1 << 15)
    ot.set_alloc_band(0.1, oracle.BAND_NORMAL_DDA)
    gt.set_alloc_band(0.1)
    gt.set_option("band_mode", 1)
    kinv = np.linalg.inv(synth.K_matrix(W, H).astype(np.float64)).astype(np.float32)
    poses = synth.camera_loop(60)
    prims = synth.room_primitives()
    for i in (0, 2, 4, 9):
        z = synth.render_room_verts(poses[i], W, H, prims)[..., 2]
        d16 = (z * 5000.0).round().clamp(0, 65535).to(torch.uint16).cuda()
        dv, dn = torch.empty((H, W, 4), device="cuda"), torch.empty((H, W, 4), device="cuda")
        vh.preprocess(d16, kinv, dv, dn)
        torch.cuda.synchronize()
        ot.integrate(poses[i], dv.cpu().numpy(), dn.cpu().numpy())
        gt.integrate(poses[i], dv, dn)
        gt.synchronize()
        _compare(ot, gt)
    assert len(gt.allocated()) > 1000


def test_tsdf_update_variants(oracle, vh, torch_cuda):
    """depth_truncation (VoxelUtils.cu:815) and weight_sample (:827): bit-exact voxels."""
    torch = torch_cuda
    kw = dict(numBuckets=1 << 14, numVoxelBlocks=4096, truncation=0.04, truncScale=0.02, integrationWeightSample=10)
    for flags in (1, 2, 3):
        ot, gt = pair(oracle, vh, "fused-ballot-walk", sem=1, overflow=False, **kw)
        ot.set_integrate_flags(flags)
        gt.set_option("depth_truncation", flags & 1)
        gt.set_option("weight_sample", (flags >> 1) & 1)
        verts = synth.sphere_inside_scene()
        d_verts = torch.from_numpy(verts).cuda()
        for _ in range(3):
            ot.integrate(I4, verts)
            gt.integrate(I4, d_verts)
        gt.synchronize()
        _compare(ot, gt)
        w = gt.sdf_blocks()["weight"]
        assert (w.max() > 25.0) == bool(flags & 2)


@pytest.mark.parametrize("variant", list(VARIANTS))
def test_ray_dda_band(oracle, vh, torch_cuda, variant):
    """VH_BAND_RAY_DDA: band allocation by the block DDA along the viewing ray (no normals), every frame form, vertex maps
    and -- the DDA needs no normal map -- the sensor-depth entry point; the room with a moving camera, slot-exact."""
    torch = torch_cuda
    W, H = 320, 240
    ot, gt = pair(oracle, vh, variant, W, H, 1, overflow=False, numBuckets=1 << 15, numVoxelBlocks=1 << 15)
    ot.set_alloc_band(0.1, oracle.BAND_RAY_DDA)
    gt.set_alloc_band(0.1)
    gt.set_option("band_mode", vh.BAND_RAY_DDA)
    kinv = np.linalg.inv(synth.K_matrix(W, H).astype(np.float64)).astype(np.float32)
    poses = synth.camera_loop(60)
    prims = synth.room_primitives()
    for n, i in enumerate((0, 2, 4, 9, 11, 30)):
        v = synth.render_room_verts(poses[i], W, H, prims).numpy()
        if n % 2:
            d16 = np.round(v[..., 2] * 5000.0).clip(0, 65535).astype(np.uint16)
            ot.integrate(poses[i], oracle.preprocess(d16, kinv)[0])
            gt.integrate_depth(poses[i], torch.from_numpy(d16).cuda(), kinv)
        else:
            ot.integrate(poses[i], v)
            gt.integrate(poses[i], torch.from_numpy(v).cuda())
        gt.synchronize()
        _compare(ot, gt)
    assert len(gt.allocated()) > 1500
    ot.close()
    gt.close()


def test_ray_dda_band_on_native_ranks(oracle, vh, torch_cuda):
    """... and through the native exchange: the key generation of vh_dist walks the same DDA (records carry frame | launch
    rank | step)."""
    torch = torch_cuda
    from test_gpu_dist_loopback import _camera_frames, _feed
    from test_sharding_cpu import check_shard_against_full
    W, H, world, batch, band = 320, 240, 2, 2, 0.1
    kw = dict(numBuckets=1 << 14, numVoxelBlocks=1 << 14)
    frames, kinv = _camera_frames(oracle, torch, world, 4, W, H, True)
    full = oracle.OracleTable(oracle.default_params(**kw), W, H, 1)
    full.set_alloc_band(band, oracle.BAND_RAY_DDA)
    g = vdist.NativeGroup(vh.default_params(**kw), W, H, 1, world, batch, sensor_k_inv=kinv, key_capacity=W * H * batch,
                          band=band, options={"band_mode": vh.BAND_RAY_DDA})
    _feed(g, full, frames, batch)
    g.flush()
    plan = vdist.ShardPlan(kw["numBuckets"], world)
    total = 0
    for r, t in enumerate(g.tables):
        total += check_shard_against_full(t, full, *plan.bucket_range(r), 5)
        assert t.counters()["bin_overflow"] == 0
    assert total == len(full.allocated()) > 500
    g.close()
    full.close()


def test_overflow_frames_choose_their_form_by_size(oracle, vh, torch_cuda):
    """Option "pipeline_overflow" 1 (the default): with the list on, a frame is one serialised launch only while few
    workgroups would have to wait and acquire (160x120: 80 claim tiles); a 640x480 frame (1 200) runs as two launches, which
    is 2.5 times faster at C2's size.  0 = never, 2 = always.  The bits are the same in every form (the tests above)."""
    torch = torch_cuda
    kw = dict(numBuckets=512, bucketSize=2, numVoxelBlocks=4096, attachedLinkedListSize=8)
    for (W, H, mode, pipelined) in ((160, 120, 1, True), (640, 480, 1, False), (640, 480, 2, True), (160, 120, 0, False)):
        gt = vh.SDFHashtable(vh.default_params(**kw), W, H, 1)
        gt.set_option("overflow_list", 1)
        gt.set_option("pipeline", 1)
        gt.set_option("pipeline_overflow", mode)
        v = torch.from_numpy(synth.render_room_verts(synth.camera_loop(40)[3], W, H, synth.room_primitives()).numpy()).cuda()
        gt.set_profiling(True)
        for _ in range(3):
            gt.integrate(synth.camera_loop(40)[3], v)
        gt.synchronize()
        kt = gt.kernel_times(reset=True)
        assert (kt["frame_pipelined_ms"] > 0) == pipelined and (kt["frame_scan_claim_ms"] > 0) == (not pipelined), (W, H, mode, kt)
        gt.close()
