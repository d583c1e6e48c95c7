"""The native multi-GPU host (include/voxelhash_dist.h: vh_dist_step_batch / vh_dist_raycast -- the code `bench.py --gpus N`
runs) with R > 1 ranks on ONE GPU, under the oracle.

RCCL refuses two ranks on one device, so the ranks of these tests are joined by the library's loop-back transport
(vh_dist_loopback_id): R vh_dist instances of this process, one host thread per rank, peer buffers copied with
hipMemcpyAsync on the calling rank's stream.  Everything else is the code the RCCL ranks run: the three buffer sets, the
generated / ready / first events, the deferred frame of pipeline_shards 2, one key bin per (owner, batch), the fixed-slot
view round.  Every shard must equal its bucket slice of ONE unsharded oracle table driven through the multi-camera frame
(dist.reference_multi_camera_frame), slot for slot and bit for bit, and every rank's raycast over the shards must equal the
oracle's raycast of that table."""
import numpy as np
import pytest

from test_gpu_configs import assert_slice_equals, compare_blocks, shard_properties
from test_sharding_cpu import check_shard_against_full
from voxelhashing_demo_amd import dist as vdist
from voxelhashing_demo_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["by size rule", "fused"])
def generation_form(request, monkeypatch):
    """Every test of this file twice: with the library's defaults -- the walk-free multi-camera frame (flatten_variant 4), whose key
    generation always runs in launches of its own -- and as bench.py's N-rank `value` runs at C5's shard sizes: the reference's walk
    (flatten_variant 3) with the key generation forced into the frame launches wherever they can carry it (VOXELHASH_DIST_FUSED=2,
    read by vh_dist_create; the size rule of option "fused_generation" 1 would keep tables as small as these on separate launches)."""
    if request.param == "fused":
        monkeypatch.setenv("VOXELHASH_DIST_FUSED", "2")
        plain = vdist.NativeDist.__init__

        def with_reference_walk(self, *a, **k):
            plain(self, *a, **k)
            self.table.set_option("flatten_variant", 3)
        monkeypatch.setattr(vdist.NativeDist, "__init__", with_reference_walk)
    else:
        monkeypatch.delenv("VOXELHASH_DIST_FUSED", raising=False)
    return request.param


def _camera_frames(oracle, torch, world, steps, W, H, sensor, loop=40, stride=3):
    """frames[s][r] = (pose, verts as the oracle sees them (numpy), device tensor the rank feeds)."""
    prims = synth.room_primitives()
    kinv = np.linalg.inv(synth.K_matrix(W, H).astype(np.float64)).astype(np.float32)
    out = []
    for s in range(steps):
        cams = []
        for r in range(world):
            pose = synth.camera_loop(loop, phase=vdist.camera_phase(r, world))[(stride * s) % loop]
            v = synth.render_room_verts(pose, W, H, prims).numpy()
            if sensor:
                d16 = np.round(v[..., 2] * 5000.0).clip(0, 65535).astype(np.uint16)
                v = oracle.preprocess(d16, kinv)[0]
                cams.append((pose, v, torch.from_numpy(d16).cuda()))
            else:
                cams.append((pose, v, torch.from_numpy(v).cuda()))
        out.append(cams)
    torch.cuda.synchronize()
    return out, kinv


def _feed(group, full, frames, batch):
    """Feeds frames[s][r] in exchanges of `batch` multi-camera frames to the group and frame by frame to the one table."""
    world = group.world
    for s0 in range(0, len(frames) - batch + 1, batch):
        chunk = frames[s0:s0 + batch]
        group.step([[chunk[b][r][0] for b in range(batch)] for r in range(world)],
                   [[chunk[b][r][2] for b in range(batch)] for r in range(world)])
        for cams in chunk:
            vdist.reference_multi_camera_frame(full, [c[0] for c in cams], [c[1] for c in cams])


@pytest.mark.parametrize("world,batch,sensor", [(2, 3, True), (2, 1, False), (4, 2, True), (3, 2, False), (8, 2, True)])
def test_native_ranks_equal_one_oracle_table(oracle, vh, torch_cuda, world, batch, sensor):
    """R ranks, several exchanges (so that all three buffer sets are reused and the deferred frame crosses exchanges), both
    packet formats, a ragged bucket split (R = 3), then a raycast round per rank -- twice, back to back."""
    torch = torch_cuda
    W, H = 320, 240
    kw = dict(numBuckets=1 << 14, numVoxelBlocks=1 << 13)
    steps = 5 * batch
    frames, kinv = _camera_frames(oracle, torch, world, steps, W, H, sensor)
    full = oracle.OracleTable(oracle.default_params(**kw), W, H, 1)
    # (the library's default bin size for every split, the ragged three-way one included)
    g = vdist.NativeGroup(vh.default_params(**kw), W, H, 1, world, batch, sensor_k_inv=kinv if sensor else None)
    assert all(nd.transport == "loopback" and nd.comm_info() == (r, world) for r, nd in enumerate(g.ranks))
    g.self_check()                      # the start-up check bench.py runs at N > 1: a pattern through both collectives
    for nd in g.ranks:
        nd.set_option("phase_timing", 1)
    _feed(g, full, frames, batch)
    g.flush()
    for nd in g.ranks:                  # per-exchange phase times (bench.py: exchange_phases_us): every exchange accounted for
        ph = nd.phase_times()
        assert ph["exchanges"] == steps // batch and ph["generate"] > 0 and ph["collectives"] > 0 and ph["apply"] > 0, ph
        nd.set_option("phase_timing", 0)
    plan = vdist.ShardPlan(kw["numBuckets"], world)
    total = 0
    for r, t in enumerate(g.tables):
        total += check_shard_against_full(t, full, *plan.bucket_range(r), 5)
        c = t.counters()
        assert c["bin_overflow"] == 0 and c["epoch"] == steps
    assert total == len(full.allocated()) > 150
    poses = [c[0] for c in frames[-1]]
    outs = [torch.empty((H, W), dtype=torch.float32, device="cuda") for _ in range(world)]
    losts = [torch.zeros(1, dtype=torch.int32, device="cuda") for _ in range(world)]
    other = [c[0] for c in frames[0]]
    for ps in (other, poses):                             # back-to-back rounds reuse the view table and the slot buffers
        g.raycast(ps, outs, 2048, losts=losts)
    g.flush()
    torch.cuda.synchronize()
    for r in range(world):
        want = full.raycast(poses[r])
        assert int(losts[r].item()) == 0
        assert np.array_equal(outs[r].cpu().numpy().view(np.uint32), want.view(np.uint32)), f"view {r}"
        assert (want > 0).mean() > 0.3
    g.close()
    full.close()


def test_native_ranks_keep_feeding_after_a_raycast(oracle, vh, torch_cuda):
    """Exchanges, a raycast round (which drains the pipeline), more exchanges: the buffer-set rotation restarts cleanly."""
    torch = torch_cuda
    W, H, world, batch = 320, 240, 2, 2
    kw = dict(numBuckets=1 << 14, numVoxelBlocks=1 << 13)
    frames, kinv = _camera_frames(oracle, torch, world, 12, W, H, True)
    full = oracle.OracleTable(oracle.default_params(**kw), W, H, 1)
    g = vdist.NativeGroup(vh.default_params(**kw), W, H, 1, world, batch, sensor_k_inv=kinv)
    outs = [torch.empty((H, W), dtype=torch.float32, device="cuda") for _ in range(world)]
    _feed(g, full, frames[:4], batch)
    g.raycast([c[0] for c in frames[3]], outs, 2048)
    torch.cuda.synchronize()
    mid = [o.cpu().numpy().copy() for o in outs]
    for r in range(world):
        assert np.array_equal(mid[r].view(np.uint32), full.raycast(frames[3][r][0]).view(np.uint32))
    _feed(g, full, frames[4:], batch)
    g.flush()
    plan = vdist.ShardPlan(kw["numBuckets"], world)
    for r, t in enumerate(g.tables):
        check_shard_against_full(t, full, *plan.bucket_range(r), 5)
    g.close()
    full.close()


def test_ranks_in_different_generation_forms(oracle, vh, torch_cuda, generation_form):
    """The form of the key generation is each rank's own decision (the size rule reads the rank's own shard, and a ragged bucket
    split can straddle it): ranks 0 and 2 forced to carry it in their frame launches, ranks 1 and 3 to launch it separately.  Both
    forms issue the same collectives in the same call, so the exchange neither hangs nor mixes frames: every shard equals its
    slice of the one oracle table, and a raycast round between such ranks equals the oracle's."""
    if generation_form != "by size rule":
        pytest.skip("the forms are set rank by rank here")
    torch = torch_cuda
    W, H, world, batch = 320, 240, 4, 2
    kw = dict(numBuckets=1 << 14, numVoxelBlocks=1 << 13)
    frames, kinv = _camera_frames(oracle, torch, world, 10, W, H, True)
    full = oracle.OracleTable(oracle.default_params(**kw), W, H, 1)
    g = vdist.NativeGroup(vh.default_params(**kw), W, H, 1, world, batch, sensor_k_inv=kinv)
    for r, nd in enumerate(g.ranks):
        nd.table.set_option("flatten_variant", 3)                 # (the walk-free multi-camera frame never carries the generation)
        nd.set_option("fused_generation", 2 if r % 2 == 0 else 0)
    _feed(g, full, frames[:6], batch)
    assert [nd.generation_form() for nd in g.ranks] == ["fused", "separate", "fused", "separate"]
    outs = [torch.empty((H, W), dtype=torch.float32, device="cuda") for _ in range(world)]
    g.raycast([c[0] for c in frames[5]], outs, 2048)
    torch.cuda.synchronize()
    for r in range(world):
        assert np.array_equal(outs[r].cpu().numpy().view(np.uint32), full.raycast(frames[5][r][0]).view(np.uint32))
    _feed(g, full, frames[6:], batch)
    g.flush()
    plan = vdist.ShardPlan(kw["numBuckets"], world)
    total = 0
    for r, t in enumerate(g.tables):
        total += check_shard_against_full(t, full, *plan.bucket_range(r), 5)
        assert t.counters()["bin_overflow"] == 0
    assert total == len(full.allocated()) > 150
    g.close()
    full.close()


def test_raycast_round_that_finds_its_own_slot_capacity(oracle, vh, torch_cuda, monkeypatch):
    """vh_dist_raycast_auto (what the C++ facade's sharded raycast calls): the first try with far too few record slots (16),
    every rank's lost count gathered, the round repeated for ALL ranks with room for what was lost, until every view is whole;
    depth and normals equal the oracle's raycast of the one table."""
    torch = torch_cuda
    W, H, world, batch = 320, 240, 2, 2
    kw = dict(numBuckets=1 << 14, numVoxelBlocks=1 << 13)
    frames, kinv = _camera_frames(oracle, torch, world, 6, W, H, True)
    full = oracle.OracleTable(oracle.default_params(**kw), W, H, 1)
    g = vdist.NativeGroup(vh.default_params(**kw), W, H, 1, world, batch, sensor_k_inv=kinv)
    _feed(g, full, frames, batch)
    g.ranks[0].set_option("raycast_auto_start", 16)              # (rank 1 proposes the default 4096 ... which is capped by the
    g.ranks[1].set_option("raycast_auto_start", 24)              #  shard's pool; the ranks start from the LARGEST proposal: 24)
    poses = [c[0] for c in frames[-1]]
    outs = [torch.empty((H, W), dtype=torch.float32, device="cuda") for _ in range(world)]
    nrm = [torch.empty((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(world)]
    caps = g._all(lambda r, nd: nd.raycast_auto(poses[r], outs[r], nrm[r]))
    assert caps[0] == caps[1] > 24                               # (the ranks agree; 24 slots were not enough)
    for r in range(world):
        od, on = full.raycast(poses[r], normals=True)
        assert np.array_equal(outs[r].cpu().numpy().view(np.uint32), od.view(np.uint32))
        assert np.array_equal(nrm[r].cpu().numpy().view(np.uint32), on.view(np.uint32))
    g.close()
    full.close()


def test_native_ranks_with_a_user_stream(oracle, vh, torch_cuda):
    """vh_dist_set_user_stream: the frames are produced on a torch stream right before the call and overwritten right after
    it, the raycast image is consumed on that stream right after the call -- no host synchronisation anywhere."""
    torch = torch_cuda
    W, H, world, batch = 320, 240, 2, 2
    kw = dict(numBuckets=1 << 14, numVoxelBlocks=1 << 13)
    frames, kinv = _camera_frames(oracle, torch, world, 8, W, H, True)
    full = oracle.OracleTable(oracle.default_params(**kw), W, H, 1)
    g = vdist.NativeGroup(vh.default_params(**kw), W, H, 1, world, batch, sensor_k_inv=kinv)
    st = torch.cuda.Stream()
    for nd in g.ranks:
        nd.order_against(st)
    staging = [[torch.empty((H, W), dtype=torch.uint16, device="cuda") for _ in range(batch)] for _ in range(world)]
    junk = torch.full((H, W), 7, dtype=torch.uint16, device="cuda")
    with torch.cuda.stream(st):
        for s0 in range(0, 8, batch):
            chunk = frames[s0:s0 + batch]
            for r in range(world):
                for b in range(batch):
                    staging[r][b].copy_(chunk[b][r][2], non_blocking=True)      # produced on the stream ...
            g.step([[chunk[b][r][0] for b in range(batch)] for r in range(world)], staging)
            for r in range(world):
                for b in range(batch):
                    staging[r][b].copy_(junk, non_blocking=True)                # ... and clobbered right behind the call
            for cams in chunk:
                vdist.reference_multi_camera_frame(full, [c[0] for c in cams], [c[1] for c in cams])
        outs = [torch.zeros((H, W), dtype=torch.float32, device="cuda") for _ in range(world)]
        g.raycast([c[0] for c in frames[-1]], outs, 2048)
        copies = [o.clone() for o in outs]                                       # read on the stream, no synchronisation
    st.synchronize()
    plan = vdist.ShardPlan(kw["numBuckets"], world)
    for r, t in enumerate(g.tables):
        check_shard_against_full(t, full, *plan.bucket_range(r), 5)
        assert np.array_equal(copies[r].cpu().numpy().view(np.uint32), full.raycast(frames[-1][r][0]).view(np.uint32))
    g.close()
    full.close()


def test_native_ranks_band_allocation(oracle, vh, torch_cuda):
    """Band allocation through the native exchange: records carry frame | launch rank | sample."""
    torch = torch_cuda
    W, H, world, batch, band = 320, 240, 2, 2, 0.15
    kw = dict(numBuckets=1 << 14, numVoxelBlocks=1 << 14)
    frames, _ = _camera_frames(oracle, torch, world, 4, W, H, False)
    full = oracle.OracleTable(oracle.default_params(**kw), W, H, 1)
    full.set_alloc_band(band)
    g = vdist.NativeGroup(vh.default_params(**kw), W, H, 1, world, batch, key_capacity=W * H * batch, band=band)
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


@pytest.mark.parametrize("world,batch", [(2, 1), (2, 3), (4, 2)])
def test_native_ranks_overflow_list(oracle, vh, torch_cuda, world, batch):
    """Bucket-range shards with the overflow list on (frames serialised inside their launch), through the native exchange;
    the raycast round's view tables carry chains of their own."""
    torch = torch_cuda
    W, H = 160, 120
    kw = dict(numBuckets=512, bucketSize=2, numVoxelBlocks=4096, attachedLinkedListSize=8)
    plan = vdist.ShardPlan(kw["numBuckets"], world)
    frames, _ = _camera_frames(oracle, torch, world, 6, W, H, False)
    full = oracle.OracleTable(oracle.default_params(**kw), W, H, 1)
    full.set_overflow(True, plan.per_shard)
    g = vdist.NativeGroup(vh.default_params(**kw), W, H, 1, world, batch, key_capacity=W * H * batch,
                          options={"overflow_list": 1, "pipeline_overflow": 2})
    _feed(g, full, frames, batch)
    g.flush()
    ftab, fvol = full.hash_table(), full.sdf_blocks()
    total = 0
    for r, t in enumerate(g.tables):
        lo, hi = plan.bucket_range(r)
        mine, want = t.hash_table(), ftab[lo * 2:hi * 2]
        assert np.array_equal(mine["pos"], want["pos"]) and np.array_equal(mine["offset"], want["offset"])
        assert np.array_equal(mine["ptr"] != -1, want["ptr"] != -1)
        for i in np.nonzero(mine["ptr"] != -1)[0][::3]:
            assert np.array_equal(t.block_voxels(int(mine["ptr"][i])).view(np.uint32),
                                  fvol[int(want["ptr"][i]):int(want["ptr"][i]) + 512].view(np.uint32))
        total += int((mine["ptr"] != -1).sum())
        assert t.counters()["bin_overflow"] == 0
    assert total == len(full.allocated()) and (ftab["offset"] != 0).sum() > 10
    poses = [c[0] for c in frames[-1]]
    outs = [torch.empty((H, W), dtype=torch.float32, device="cuda") for _ in range(world)]
    g.raycast(poses, outs, 1024 // world)                 # (a view table lists one imported record per entry: 512 x 2 of them)
    g.flush()
    torch.cuda.synchronize()
    for r in range(world):
        assert np.array_equal(outs[r].cpu().numpy().view(np.uint32), full.raycast(poses[r]).view(np.uint32)), r
    g.close()
    full.close()


def test_native_exchange_across_the_epoch_wrap(oracle, vh, torch_cuda, generation_form):
    """(Run in the fused form only -- the reference's walk with the key generation inside the frame launches, the lag-2 exchange: the
    form with the most frames in flight at a wrap; the library's defaults cross the wraps in tests/test_gpu_sharding.py.)
    The claim words carry a 9-bit lock epoch; at the wrap they are cleared, which the frame whose deferred half is still
    pending must not see: 1 040 multi-camera frames over two ranks (batches of 8) straddle the wraps at 511 and 1 022."""
    if generation_form != "fused":
        pytest.skip("once is enough for 1 040 frames")
    torch = torch_cuda
    kw = dict(numBuckets=1 << 10, numVoxelBlocks=4096)
    w, h, world = 64, 48, 2
    prims = synth.room_primitives()
    loops = [synth.camera_loop(40, phase=vdist.camera_phase(r, world)) for r in range(world)]
    verts = [[synth.render_room_verts(p, w, h, prims).numpy() for p in loops[r]] for r in range(world)]
    dv = [[torch.from_numpy(v).cuda() for v in verts[r]] for r in range(world)]
    torch.cuda.synchronize()
    g = vdist.NativeGroup(vh.default_params(**kw), w, h, 1, world, 8, key_capacity=w * h * 4)
    full = oracle.OracleTable(oracle.default_params(**kw), w, h, 1)
    for step in range(130):
        ks = [(8 * step + b) % 40 for b in range(8)]
        g.step([[loops[r][k] for k in ks] for r in range(world)], [[dv[r][k] for k in ks] for r in range(world)])
        for k in ks:
            vdist.reference_multi_camera_frame(full, [loops[r][k] for r in range(world)], [verts[r][k] for r in range(world)])
    g.flush()
    plan = vdist.ShardPlan(kw["numBuckets"], world)
    total = 0
    for r, t in enumerate(g.tables):
        c = t.counters()
        assert c["epoch"] == 1040 and c["heap_exhausted"] == 0 and c["bin_overflow"] == 0
        total += check_shard_against_full(t, full, *plan.bucket_range(r), 5)
    assert total > 20
    g.close()
    full.close()


def test_a_rank_that_never_arrives_fails_the_collective(vh, torch_cuda, monkeypatch):
    """One host thread driving two ranks in turn cannot work (the first call waits for the second rank): the transport
    says so instead of hanging.  Membership of a group is checked at creation."""
    torch = torch_cuda
    from voxelhashing_demo_amd import _lib as L
    W, H = 64, 48
    kw = dict(numBuckets=1 << 10, numVoxelBlocks=512)
    uid = vdist.loopback_id()
    nd = vdist.NativeDist(vh.default_params(**kw), W, H, 1, 0, 2, 1, uid)
    monkeypatch.setenv("VOXELHASH_LOOPBACK_TIMEOUT_S", "1")
    frame = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    with pytest.raises(L.VoxelHashError, match="did not arrive"):
        nd.step([np.eye(4, dtype=np.float32)], [frame])                           # rank 1 never calls
    with pytest.raises(L.VoxelHashError):
        vdist.NativeDist(vh.default_params(**kw), W, H, 1, 0, 2, 1, uid)          # the same rank twice
    with pytest.raises(L.VoxelHashError):
        vdist.NativeDist(vh.default_params(**kw), W, H, 1, 1, 3, 1, uid)          # another world size
    nd.close()
    with pytest.raises(L.VoxelHashError):
        vdist.NativeDist(vh.default_params(**kw), W, H, 1, 0, 2, 1, uid)          # the group is gone with its last rank


# ---------------------------------------------------------------------------------------------
# BASELINE.json configs[3] and configs[4] at their sizes, through the native exchange
# ---------------------------------------------------------------------------------------------
def _config_frames(vh, torch, world, steps, W, H, frames_on_loop=500, stride=5):
    prims = synth.room_primitives()
    kinv = np.linalg.inv(synth.K_matrix(W, H).astype(np.float64)).astype(np.float32)
    out = []
    for s in range(steps):
        cams = []
        for r in range(world):
            pose = synth.camera_loop(frames_on_loop, phase=vdist.camera_phase(r, world))[(stride * s) % frames_on_loop]
            dv = synth.render_room_verts(pose, W, H, prims, device="cuda")
            d16 = (dv[..., 2] * 5000.0).round().clamp(0, 65535).to(torch.uint16)
            vh.preprocess(d16, kinv, dv, torch.empty_like(dv))          # the vertex map preProcess makes of the image
            torch.cuda.synchronize()
            cams.append((pose, dv.cpu().numpy(), d16))
        out.append(cams)
    return out, kinv


def _check_config(g, full, plan, kw_rank, block_stride):
    otab = full.hash_table()
    total = 0
    for r, t in enumerate(g.tables):
        lo, hi = plan.bucket_range(r)
        gtab = t.hash_table()
        assert_slice_equals(gtab, otab, lo, hi, 5, f"shard {r}")
        compare_blocks(t, gtab, full, otab[lo * 5:hi * 5], every=block_stride)
        total += int((gtab["ptr"] != -1).sum())
        shard_properties(t, kw_rank["numVoxelBlocks"])
    assert total == len(full.allocated())
    return total


def test_c4_four_cameras_four_native_ranks(oracle, vh, torch_cuda):
    """configs[3]: 4 virtual 640x480 cameras, 2^20 buckets over 4 ranks, 2^18 blocks per rank, 10 multi-camera frames in
    exchanges of 2, sensor-depth packets, the library's default bin size."""
    torch = torch_cuda
    W, H, world, batch = 640, 480, 4, 2
    kw_rank = dict(numBuckets=1 << 20, numVoxelBlocks=1 << 18)
    kw_full = dict(numBuckets=1 << 20, numVoxelBlocks=1 << 16)
    frames, kinv = _config_frames(vh, torch, world, 10, W, H)
    full = oracle.OracleTable(oracle.default_params(**kw_full), W, H, 1)
    g = vdist.NativeGroup(vh.default_params(**kw_rank), W, H, 1, world, batch, sensor_k_inv=kinv)
    _feed(g, full, frames, batch)
    g.flush()
    plan = vdist.ShardPlan(kw_full["numBuckets"], world)
    assert _check_config(g, full, plan, kw_rank, 1) > 2000
    poses = [c[0] for c in frames[-1]]
    outs = [torch.empty((H, W), dtype=torch.float32, device="cuda") for _ in range(world)]
    losts = [torch.zeros(1, dtype=torch.int32, device="cuda") for _ in range(world)]
    g.raycast(poses, outs, 8192, losts=losts)
    g.flush()
    torch.cuda.synchronize()
    for r in range(world):
        ref = full.raycast(poses[r])
        assert int(losts[r].item()) == 0 and (ref > 0).mean() > 0.5
        assert np.array_equal(outs[r].cpu().numpy().view(np.uint32), ref.view(np.uint32)), f"view {r}"
    g.close()
    full.close()


def test_c5_eight_streams_eight_native_ranks(oracle, vh, torch_cuda):
    """configs[4]: 8 x 1920x1080 streams, 2^24 buckets (1.68 GB of VoxelEntry) over 8 ranks, 1 cm voxels; the voxel pool is
    capped at 2^16 blocks per rank (the config's 2^21 is a capacity; C3 runs a 2^21 pool).  Two exchanges of one
    multi-camera frame each against the oracle, then the same frame again until every shard's set stops growing."""
    torch = torch_cuda
    W, H, world = 1920, 1080, 8
    kw_rank = dict(numBuckets=1 << 24, numVoxelBlocks=1 << 16, voxelSize=0.01)
    kw_full = dict(numBuckets=1 << 24, numVoxelBlocks=1 << 18, voxelSize=0.01)
    frames, kinv = _config_frames(vh, torch, world, 2, W, H)
    full = oracle.OracleTable(oracle.default_params(**kw_full), W, H, 1)
    g = vdist.NativeGroup(vh.default_params(**kw_rank), W, H, 1, world, 1, sensor_k_inv=kinv)
    _feed(g, full, frames, 1)
    g.flush()
    plan = vdist.ShardPlan(kw_full["numBuckets"], world)
    assert _check_config(g, full, plan, kw_rank, 5) > 5000
    poses = [c[0] for c in frames[-1]]
    outs = [torch.empty((H, W), dtype=torch.float32, device="cuda") for _ in range(world)]
    losts = [torch.zeros(1, dtype=torch.int32, device="cuda") for _ in range(world)]
    g.raycast(poses, outs, 16384, losts=losts)
    g.flush()
    torch.cuda.synchronize()
    for r in range(2):
        ref = full.raycast(poses[r])
        assert int(losts[r].item()) == 0 and (ref > 0).mean() > 0.5
        assert np.array_equal(outs[r].cpu().numpy().view(np.uint32), ref.view(np.uint32)), f"view {r}"
    full.close()
    prev = -1
    last = frames[-1]
    for _ in range(10):
        g.step([[last[r][0]] for r in range(world)], [[last[r][2]] for r in range(world)])
        g.flush()
        cur = sum(t.counters()["allocated_total"] for t in g.tables)
        if cur == prev:
            break
        prev = cur
    assert cur == prev
    for t in g.tables:
        shard_properties(t, kw_rank["numVoxelBlocks"])
    g.close()
