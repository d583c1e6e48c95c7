"""Pipelined frames (option "pipeline", vh_integrate_batch): one launch per frame -- the commit and TSDF
update of frame i ride in the launch of frame i+1 -- must leave exactly what the two-launch frames
leave: every test compares with the oracle slot for slot and bit for bit."""
import numpy as np
import pytest

from conftest import entries_as_set
from test_gpu_parity import _compare
from voxelhashing_demo_amd import synth

pytestmark = pytest.mark.gpu
I4 = np.eye(4, dtype=np.float32)


def room_frames(torch, W, H, idx, loop=500):
    poses = synth.camera_loop(loop)
    prims = synth.room_primitives()
    return [(poses[i], synth.render_room_verts(poses[i], W, H, prims).numpy()) for i in idx]


@pytest.mark.parametrize("sem", [0, 1])
@pytest.mark.parametrize("chunk", [1, 2, 5])
def test_batches_equal_oracle_frames(oracle, vh, torch_cuda, sem, chunk):
    _batches_equal_oracle_frames(oracle, vh, torch_cuda, sem, chunk, 0)


def test_batches_with_non_temporal_walk(oracle, vh, torch_cuda):
    """"walk_nt" (what tables beyond the Infinity Cache get by default) on a small table."""
    _batches_equal_oracle_frames(oracle, vh, torch_cuda, 1, 2, 1)


@pytest.mark.parametrize("chunk", [1, 3])
def test_batches_with_the_occupancy_index_walk(oracle, vh, torch_cuda, chunk):
    """flatten_variant 4 under pipelining: the walk over the bucket-occupancy bitmap skips the slot the
    concurrent commit phase is filling, like the walk over the entries does."""
    _batches_equal_oracle_frames(oracle, vh, torch_cuda, 1, chunk, 0, walk=4)


@pytest.mark.parametrize("walk_nt", [0, 1])
def test_walk_free_frame_with_a_launch_tile_per_wave(oracle, vh, torch_cuda, walk_nt):
    """The walk-free frame's claim role with a launch tile per WAVE (claim_tile_wave; lean builds 7 / 8): what images of more
    than 2400 launch tiles get -- here 1024x640 = 2560 tiles.  The same keys with the same ranks must reach the probes, so the
    table is the oracle's slot for slot."""
    _batches_equal_oracle_frames(oracle, vh, torch_cuda, 1, 3, walk_nt, walk=4, size=(1024, 640))


def _batches_equal_oracle_frames(oracle, vh, torch_cuda, sem, chunk, walk_nt, walk=3, size=(640, 480)):
    """The sphere scene twice (frame 1 demands keys frame 0 is still inserting), then a moving camera:
    checked after every batch, whatever the batch length."""
    torch = torch_cuda
    W, H = size
    kw = dict(numBuckets=1 << 15, numVoxelBlocks=1 << 13)
    ot = oracle.OracleTable(oracle.default_params(**kw), W, H, sem)
    gt = vh.SDFHashtable(vh.default_params(**kw), W, H, sem)
    gt.set_option("walk_nt", walk_nt)
    gt.set_option("flatten_variant", walk)
    sphere = synth.sphere_inside_scene(W, H)
    frames = [(I4, sphere)] * 3 + room_frames(torch, W, H, (0, 1, 2, 3, 8, 9, 10))
    for s in range(0, len(frames), chunk):
        part = frames[s:s + chunk]
        d = [torch.from_numpy(np.ascontiguousarray(v)).cuda() for _, v in part]
        gt.integrate_batch([p for p, _ in part], d)
        for p, v in part:
            ot.integrate(p, v)
        _compare(ot, gt)
    assert len(gt.allocated()) > 300


def test_collision_stress_pipelined(oracle, vh, torch_cuda):
    """G5: 64 buckets x 2 -- heavy bucket contention while insertions are in flight: the claim phase of
    frame i+1 must see each bucket exactly as commit(i) leaves it."""
    torch = torch_cuda
    kw = dict(numBuckets=64, bucketSize=2, numVoxelBlocks=1024)
    for batch in (2, 9):
        ot = oracle.OracleTable(oracle.default_params(**kw), 640, 480, 0)
        gt = vh.SDFHashtable(vh.default_params(**kw), 640, 480, 0)
        verts = synth.sphere_inside_scene()
        d = torch.from_numpy(verts).cuda()
        for _ in range(3):
            gt.integrate_batch([I4] * batch, [d] * batch)
            for _ in range(batch):
                ot.integrate(I4, verts)
            _compare(ot, gt)
        assert len(gt.allocated()) == 101


@pytest.mark.parametrize("bucket_size", [10, 16])
def test_wide_buckets_run_pipelined(oracle, vh, torch_cuda, bucket_size):
    """The claim word names the slot an insertion in flight takes with 4 bits: buckets of up to 16 slots are
    pipelined (one launch per frame) like the 5-slot ones.  Few buckets, so that insertions land deep in them."""
    torch = torch_cuda
    kw = dict(numBuckets=32, bucketSize=bucket_size, numVoxelBlocks=1024)
    ot = oracle.OracleTable(oracle.default_params(**kw), 640, 480, 0)
    gt = vh.SDFHashtable(vh.default_params(**kw), 640, 480, 0)
    gt.set_profiling(True)
    verts = synth.sphere_inside_scene()
    d = torch.from_numpy(verts).cuda()
    for _ in range(4):
        gt.integrate_batch([I4] * 5, [d] * 5)
        for _ in range(5):
            ot.integrate(I4, verts)
        _compare(ot, gt)
    times = gt.kernel_times()
    assert times["frame_pipelined_ms"] > 0.0 and times["frame_commit_integrate_ms"] == 0.0, times
    slots = np.flatnonzero(gt.hash_table()["ptr"] != -1) % bucket_size          # (VH_FREE_BLOCK)
    assert slots.max() >= 8                                 # (entries beyond slot 7: the fourth bit is in use)


def test_pipeline_option_streaming_and_reused_buffer(oracle, vh, torch_cuda):
    """option "pipeline": plain vh_integrate / vh_integrate_depth calls, ONE device buffer overwritten for
    every frame (the deferred half works from a private copy), observers flush on their own."""
    torch = torch_cuda
    W, H = 320, 240
    kw = dict(numBuckets=1 << 14, numVoxelBlocks=1 << 13)
    ot = oracle.OracleTable(oracle.default_params(**kw), W, H, 1)
    gt = vh.SDFHashtable(vh.default_params(**kw), W, H, 1)
    gt.set_option("pipeline", 1)
    kinv = np.linalg.inv(synth.K_matrix(W, H).astype(np.float64)).astype(np.float32)
    buf = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
    buf16 = torch.empty((H, W), dtype=torch.uint16, device="cuda")
    depth = torch.empty((H, W), dtype=torch.float32, device="cuda")
    for n, (pose, v) in enumerate(room_frames(torch, W, H, range(0, 36, 3))):
        if n % 3 == 2:                                     # a sensor frame in between: uint16 image, vertices in place
            d16 = np.round(v[..., 2] * 5000.0).clip(0, 65535).astype(np.uint16)
            buf16.copy_(torch.from_numpy(d16))
            torch.cuda.synchronize()
            gt.integrate_depth(pose, buf16, kinv)
            ot.integrate(pose, oracle.preprocess(d16, kinv)[0])
        else:
            buf.copy_(torch.from_numpy(v))
            torch.cuda.synchronize()
            gt.integrate(pose, buf)
            ot.integrate(pose, v)
        buf.zero_()                                        # the caller's buffers are free again
        buf16.zero_()
        torch.cuda.synchronize()
        if n % 4 == 3:                                     # observers see completed frames
            assert gt.counters()["occupied"] == ot.compact_count()
            gt.raycast(pose, depth)
            gt.synchronize()
            assert np.array_equal(depth.cpu().numpy().view(np.uint32), ot.raycast(pose).view(np.uint32))
    _compare(ot, gt)
    # collection and deletion in the middle of a pipelined run
    pose, v = room_frames(torch, W, H, (40,))[0]
    buf.copy_(torch.from_numpy(v))
    gt.integrate(pose, buf)
    ot.integrate(pose, v)
    a = ot.garbage_collect(0.05)
    gt.garbage_collect(0.05)
    assert gt.counters()["last_freed"] == a
    gt.integrate(pose, buf)
    ot.integrate(pose, v)
    gt.set_option("pipeline", 0)                           # (flushes)
    gt.integrate(pose, buf)
    ot.integrate(pose, v)
    _compare(ot, gt)


def test_band_allocation_pipelined(oracle, vh, torch_cuda):
    torch = torch_cuda
    W, H = 320, 240
    kw = dict(numBuckets=1 << 14, numVoxelBlocks=1 << 14)
    ot = oracle.OracleTable(oracle.default_params(**kw), W, H, 1)
    gt = vh.SDFHashtable(vh.default_params(**kw), W, H, 1)
    ot.set_alloc_band(0.15)
    gt.set_alloc_band(0.15)
    frames = room_frames(torch, W, H, (0, 2, 4, 6, 8))
    gt.integrate_batch([p for p, _ in frames], [torch.from_numpy(v).cuda() for _, v in frames])
    for p, v in frames:
        ot.integrate(p, v)
    _compare(ot, gt)
    assert len(gt.allocated()) > 500


@pytest.mark.parametrize("band", [0.0, 0.15])
def test_lean_builds_of_the_pipelined_launch_switched_in_a_run(oracle, vh, torch_cuda, band):
    """The pipelined launch has builds with the context's option flags folded in (one per flag set that has such a build; every
    other flag set -- overflow list, TSDF-update variants, the other bands: tests/test_gpu_overflow.py -- runs the generic
    build): the same frames through the build without and the build with non-temporal walk loads, switched in the middle of a run."""
    torch = torch_cuda
    W, H = 320, 240
    kw = dict(numBuckets=1 << 14, numVoxelBlocks=1 << 14)
    ot = oracle.OracleTable(oracle.default_params(**kw), W, H, 1)
    gt = vh.SDFHashtable(vh.default_params(**kw), W, H, 1)
    gt.set_option("pipeline", 1)
    ot.set_alloc_band(band)
    gt.set_alloc_band(band)
    for i, (p, v) in enumerate(room_frames(torch, W, H, (0, 1, 2, 3, 4, 5, 6, 7))):
        gt.set_option("walk_nt", 1 if i in (2, 3, 6) else 0)             # (a change of option flushes the pending half first)
        gt.integrate(p, torch.from_numpy(v).cuda())
        ot.integrate(p, v)
        if i in (3, 7):
            _compare(ot, gt)
    assert len(gt.allocated()) > 200


def test_band_switched_between_pipelined_frames(oracle, vh, torch_cuda):
    """The pipelined launch comes in two builds, with and without the band code; which one runs follows the NEW frame's band,
    while the half it carries for the pending frame does not depend on it: band on and off from frame to frame."""
    torch = torch_cuda
    W, H = 320, 240
    kw = dict(numBuckets=1 << 14, numVoxelBlocks=1 << 14)
    ot = oracle.OracleTable(oracle.default_params(**kw), W, H, 1)
    gt = vh.SDFHashtable(vh.default_params(**kw), W, H, 1)
    gt.set_option("pipeline", 1)
    frames = room_frames(torch, W, H, (0, 1, 2, 3, 4, 5, 6))
    for i, (p, v) in enumerate(frames):
        band = 0.15 if i in (1, 2, 5) else 0.0
        ot.set_alloc_band(band)
        gt.set_alloc_band(band)                             # (does not flush: frame i-1's half rides in frame i's launch)
        gt.integrate(p, torch.from_numpy(v).cuda())
        ot.integrate(p, v)
        if i in (2, 6):
            _compare(ot, gt)
    assert len(gt.allocated()) > 500


def test_heap_shortage_refuses_whole_frames(oracle, vh, torch_cuda):
    """The documented difference: a pipelined frame whose new blocks outnumber the free blocks allocates
    none of them.  Until then the batch equals the oracle; afterwards the model stays consistent, and
    frames fit again once blocks have been freed."""
    torch = torch_cuda
    sphere = synth.sphere_inside_scene()
    probe = oracle.OracleTable(oracle.default_params(numBuckets=1 << 12, numVoxelBlocks=1024), 640, 480, 1)
    probe.integrate(I4, sphere)
    n0 = len(probe.allocated())                            # blocks frame 0 inserts
    probe.integrate(I4, sphere)
    more = len(probe.allocated()) - n0                     # ... and frame 1 would
    assert n0 > 100 and more > 10
    kw = dict(numBuckets=1 << 12, numVoxelBlocks=n0 + 3)
    gt = vh.SDFHashtable(vh.default_params(**kw), 640, 480, 1)
    ot = oracle.OracleTable(oracle.default_params(**kw), 640, 480, 1)
    d = torch.from_numpy(sphere).cuda()
    gt.integrate_batch([I4], [d])                          # frame 0: n0 new blocks, n0 + 3 free: served
    ot.integrate(I4, sphere)
    _compare(ot, gt)
    assert len(gt.allocated()) == n0
    gt.integrate_batch([I4] * 3, [d] * 3)                  # `more` wanted, 3 free: refused, frame after frame
    c = gt.counters()
    assert len(gt.allocated()) == n0 and c["heap_counter"] == 2 and c["heap_exhausted"] == 3 * more
    tab = gt.hash_table()
    alloc = tab[tab["ptr"] != -1]
    assert len(entries_as_set(alloc)) == n0 and len(set(alloc["ptr"].tolist())) == n0
    for b in range(0, len(tab), 5):                        # entries still form a prefix of every bucket
        live = tab["ptr"][b:b + 5] != -1
        assert not np.any(live[1:] & ~live[:-1])
    # the voxels kept being updated (4 frames so far) although nothing was inserted
    w = gt.sdf_blocks()["weight"]
    assert np.isclose(w.max(), 0.4, atol=1e-6)
    # free the blocks: a frame that fits is served again (the model as after frame 0), the next one refused again
    doomed = np.zeros((n0, 4), np.int32)
    doomed[:, :3] = alloc["pos"]
    gt.delete_blocks(torch.from_numpy(doomed).cuda())
    assert len(gt.allocated()) == 0
    gt.integrate_batch([I4] * 2, [d] * 2)
    c2 = gt.counters()
    assert len(gt.allocated()) == n0 and c2["heap_exhausted"] == c["heap_exhausted"] + more
    assert entries_as_set(gt.allocated()) == entries_as_set(ot.allocated())


@pytest.mark.parametrize("sem", [0, 1])
def test_ragged_empty_and_hostile_frames_pipelined(oracle, vh, torch_cuda, sem):
    """One pipelined batch through the awkward inputs of the unpipelined suite: an image that is no multiple
    of the 16x16 launch tile, an all-invalid frame between real ones (its deferred half has nothing to
    commit or update), holes, and NaN / inf / huge / denormal vertex components under a non-rigid pose."""
    torch = torch_cuda
    W, H = 200, 150
    kw = dict(numBuckets=1 << 12, numVoxelBlocks=8192)
    ot = oracle.OracleTable(oracle.default_params(**kw), W, H, sem)
    gt = vh.SDFHashtable(vh.default_params(**kw), W, H, sem)
    empty = np.zeros((H, W, 4), np.float32)
    empty[..., 3] = 1.0
    holes = synth.sphere_inside_scene(W, H)
    holes[::7, ::5, 2] = 0.0
    hostile = synth.sphere_inside_scene(W, H)
    rng = np.random.RandomState(5)
    specials = np.array([np.nan, np.inf, -np.inf, 1e-42, -1e-42, 3e38, -3e38, 1e9, -1e9, 4.3e7, -4.3e7, 1e-7, -0.0,
                         2147483.6, -2147483.6], np.float32)
    for comp in range(4):
        ys, xs = rng.randint(0, H, 300), rng.randint(0, W, 300)
        hostile[ys, xs, comp] = specials[rng.randint(0, len(specials), 300)]
    pose = np.array([[1.1, 0.05, 0, 0.1], [0, 0.9, 0.1, -0.05], [0.02, 0, 1.0, 0.2], [0, 0, 0, 1]], np.float32)
    frames = [(I4, empty), (I4, holes), (I4, empty), (I4, empty), (pose, hostile), (I4, holes), (I4, hostile), (I4, empty)]
    d = [torch.from_numpy(np.ascontiguousarray(v)).cuda() for _, v in frames]
    gt.integrate_batch([p for p, _ in frames[:5]], d[:5])
    for p, v in frames[:5]:
        ot.integrate(p, v)
    _compare(ot, gt)
    gt.set_option("pipeline", 1)
    for (p, v), dv in zip(frames[5:], d[5:]):
        gt.integrate(p, dv)
        ot.integrate(p, v)
    _compare(ot, gt)                      # (reading the model flushes the empty last frame)
    assert len(gt.allocated()) > 20 and gt.counters()["heap_exhausted"] == 0


@pytest.mark.parametrize("pipelined", [0, 1])
def test_lock_epoch_wrap(oracle, vh, torch_cuda, pipelined):
    """The claim words carry a 9-bit lock epoch: after 511 epochs the claim arrays are cleared and the
    epoch starts over (vh_reset_mutexes).  2 100 frames with collections that empty the model just before,
    on and after the wraps, so that insertions happen in the epochs around them; pipelined frames have
    both claim buffers in use when the wrap comes."""
    torch = torch_cuda
    W, H = 160, 120
    kw = dict(numBuckets=1 << 10, numVoxelBlocks=2048)
    ot = oracle.OracleTable(oracle.default_params(**kw), W, H, 1)
    gt = vh.SDFHashtable(vh.default_params(**kw), W, H, 1)
    gt.set_option("pipeline", pipelined)
    frames = room_frames(torch, W, H, (0, 5, 10, 15))
    dv = [torch.from_numpy(np.ascontiguousarray(v)).cuda() for _, v in frames]
    collect_at = {500, 506, 508, 510, 512, 515, 1010, 1015, 1019, 1021, 1023, 1026, 2030, 2036, 2040, 2044, 2047, 2050}
    check_at = {507, 513, 520, 1018, 1024, 1030, 2038, 2046, 2052, 2099}
    inserted_near_wrap = 0
    for f in range(2100):
        k = f % 4
        gt.integrate(frames[k][0], dv[k])
        ot.integrate(frames[k][0], frames[k][1])
        if f in collect_at:                                  # (a collection takes a lock epoch of its own)
            freed = ot.garbage_collect(0.0)
            gt.garbage_collect(0.0)
            assert gt.counters()["last_freed"] == freed
            inserted_near_wrap += freed
        if f in check_at:
            _compare(ot, gt)
    assert inserted_near_wrap > 200
    _compare(ot, gt)
