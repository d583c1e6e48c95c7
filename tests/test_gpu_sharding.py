"""Bucket-range shards on the GPU: R shard contexts on one device with the in-process
exchange (the RCCL transport itself is covered by the gloo test and runs on the 8-GPU node);
each shard must equal its slice of ONE unsharded oracle table after the same multi-camera frames."""
import numpy as np
import pytest

from test_sharding_cpu import _free_port, check_shard_against_full
from voxelhashing_demo_amd import dist as vdist
from voxelhashing_demo_amd import synth

pytestmark = pytest.mark.gpu

W, H = 320, 240
KW = dict(numBuckets=1 << 14, numVoxelBlocks=4096)


def cameras(world, step):
    prims = synth.room_primitives()
    out = []
    for r in range(world):
        pose = synth.camera_loop(60, phase=vdist.camera_phase(r, world))[(5 * step) % 60]
        out.append((pose, synth.render_room_verts(pose, W, H, prims).numpy()))
    return out


@pytest.mark.parametrize("world,batch", [(1, 1), (2, 1), (4, 1), (2, 2), (4, 3)])
@pytest.mark.parametrize("sem", [0, 1])
@pytest.mark.parametrize("calls", ["batched", "stepwise"])
def test_hip_shards_equal_one_oracle_table(oracle, vh, torch_cuda, world, batch, sem, calls):
    """batched: vh_generate_keys_batch + vh_apply_frames_batch (fused two-launch multi-camera
    frames); stepwise: vh_generate_keys / vh_insert_bins / vh_integrate_packets per frame."""
    torch = torch_cuda
    plan = vdist.ShardPlan(KW["numBuckets"], world)
    shards = [vdist.HipShard(vh.default_params(**KW), W, H, sem, plan, r, W * H // 4, batch=batch,
                             batched_calls=(calls == "batched"))
              for r in range(world)]
    full = oracle.OracleTable(oracle.default_params(**KW), W, H, sem)
    for step in range(0, 3, batch):
        frames = [cameras(world, step + b) for b in range(batch)]                 # frames[b][r]
        vdist.loopback_step(shards, [[frames[b][r][0] for b in range(batch)] for r in range(world)],
                            [[torch.from_numpy(frames[b][r][1]).cuda() for b in range(batch)] for r in range(world)])
        for sh in shards:
            sh.table.synchronize()
        for cams in frames:
            vdist.reference_multi_camera_frame(full, [c[0] for c in cams], [c[1] for c in cams])
    total = 0
    for r, sh in enumerate(shards):
        lo, hi = plan.bucket_range(r)
        total += check_shard_against_full(sh.table, full, lo, hi, 5)
        c = sh.table.counters()
        assert c["bin_overflow"] == 0 and c["heap_exhausted"] == 0
    assert total == len(full.allocated())
    if sem == 1:
        assert total > 100
    for sh in shards:
        sh.table.close()


def test_unsharded_gpu_multi_camera_frame_by_steps(oracle, vh, torch_cuda):
    """The same multi-camera frame through the step-level entry points of one full table."""
    torch = torch_cuda
    gt = vh.SDFHashtable(vh.default_params(**KW), W, H, 1)
    full = oracle.OracleTable(oracle.default_params(**KW), W, H, 1)
    for step in range(2):
        cams = cameras(3, step)
        vdist.reference_multi_camera_frame(gt, [c[0] for c in cams], [torch.from_numpy(c[1]).cuda() for c in cams])
        gt.synchronize()
        vdist.reference_multi_camera_frame(full, [c[0] for c in cams], [c[1] for c in cams])
    assert check_shard_against_full(gt, full, 0, KW["numBuckets"], 5) > 100


def test_key_bin_overflow_is_reported(vh, torch_cuda):
    torch = torch_cuda
    plan = vdist.ShardPlan(KW["numBuckets"], 1)
    sh = vdist.HipShard(vh.default_params(**KW), W, H, 1, plan, 0, 16)      # 15 keys per bin
    pose, verts = cameras(1, 0)[0]
    vdist.loopback_step([sh], [[pose]], [[torch.from_numpy(verts).cuda()]])
    sh.table.synchronize()
    c = sh.table.counters()
    assert c["bin_overflow"] == 1
    assert 0 < c["allocated_total"] <= 15


@pytest.mark.parametrize("sensor", [False, True])
def test_pipelined_steps_equal_sequential_over_nccl(oracle, vh, torch_cuda, sensor):
    """RCCL transport with one rank: the two-stream pipeline (generate + collectives of step i+1
    under the table work of step i) leaves exactly the table the plain step sequence leaves,
    and both equal the oracle."""
    import os

    import torch.distributed as dist
    torch = torch_cuda
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(_free_port())        # a fixed port may still be held by an earlier run
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        plan = vdist.ShardPlan(KW["numBuckets"], 1)
        transport = vdist.TorchDistTransport()
        batch, steps = 2, 5
        frames = [[cameras(1, s * batch + b)[0] for b in range(batch)] for s in range(steps)]
        kinv = np.linalg.inv(synth.K_matrix(W, H).astype(np.float64)).astype(np.float32)
        depths = None
        if sensor:        # sensor-depth packets: quantised depth, vertex maps by preProcess, uint16 images on the wire
            d16 = [[np.round(v[..., 2] * 5000.0).clip(0, 65535).astype(np.uint16) for _, v in fs] for fs in frames]
            frames = [[(p, oracle.preprocess(d, kinv)[0]) for (p, _), d in zip(fs, ds)] for fs, ds in zip(frames, d16)]
            depths = [[torch.from_numpy(d).cuda() for d in ds] for ds in d16]
        d_frames = [[(p, torch.from_numpy(v).cuda()) for p, v in fs] for fs in frames]
        tables = []
        for pipelined in (False, True):
            table_stream, front = torch.cuda.Stream(), torch.cuda.Stream()
            sh = vdist.HipShard(vh.default_params(**KW), W, H, 1, plan, 0, W * H // 4, batch=batch,
                                stream=table_stream, sets=2, sensor_k_inv=kinv if sensor else None)
            if pipelined:
                pipe = vdist.ShardedPipeline(sh, transport, table_stream, front)
                for i, fs in enumerate(d_frames):
                    pipe.feed([f[0] for f in fs], [f[1] for f in fs], depths[i] if sensor else None)
                pipe.flush()
            else:
                with torch.cuda.stream(table_stream):
                    for i, fs in enumerate(d_frames):
                        vdist.sharded_step(sh, transport, [f[0] for f in fs], [f[1] for f in fs],
                                           depths[i] if sensor else None)
                sh.table.synchronize()
            torch.cuda.synchronize()
            tables.append(sh)
        full = oracle.OracleTable(oracle.default_params(**KW), W, H, 1)
        for fs in frames:
            for pose, verts in fs:
                full.integrate(pose, verts)          # one camera: the multi-camera frame is integrate()
        for sh in tables:
            assert check_shard_against_full(sh.table, full, 0, KW["numBuckets"], 5) > 100
            sh.table.close()
    finally:
        dist.destroy_process_group()


def test_sharded_band_allocation(oracle, vh, torch_cuda):
    """Band allocation through the key exchange: ranks carry camera | launch rank | sample."""
    torch = torch_cuda
    world, batch, band = 2, 2, 0.15
    plan = vdist.ShardPlan(KW["numBuckets"], world)
    kw = dict(numBuckets=KW["numBuckets"], numVoxelBlocks=1 << 14)
    shards = [vdist.HipShard(vh.default_params(**kw), W, H, 1, plan, r, W * H, batch=batch) for r in range(world)]
    full = oracle.OracleTable(oracle.default_params(**kw), W, H, 1)
    full.set_alloc_band(band)
    for sh in shards:
        sh.table.set_alloc_band(band)
    for step in range(0, 4, batch):
        frames = [cameras(world, step + b) for b in range(batch)]
        vdist.loopback_step(shards, [[frames[b][r][0] for b in range(batch)] for r in range(world)],
                            [[torch.from_numpy(frames[b][r][1]).cuda() for b in range(batch)] for r in range(world)])
        for cams in frames:
            vdist.reference_multi_camera_frame(full, [c[0] for c in cams], [c[1] for c in cams])
    total = 0
    for r, sh in enumerate(shards):
        sh.table.synchronize()
        total += check_shard_against_full(sh.table, full, *plan.bucket_range(r), 5)
        assert sh.table.counters()["bin_overflow"] == 0
    assert total == len(full.allocated()) > 500


@pytest.mark.parametrize("world,batch", [(2, 1), (4, 2)])
@pytest.mark.parametrize("sem", [0, 1])
@pytest.mark.parametrize("calls", ["batched", "stepwise"])
def test_sensor_depth_packets(oracle, vh, torch_cuda, world, batch, sem, calls):
    """VH_PACKET_U16: the packets carry the uint16 sensor image (half the bytes of the float camera-z
    plane) and the owner recomputes z like preProcess.  Vertex maps come from preProcess on the same
    images; the oracle side keeps float packets.  Shards must equal the one oracle table bit for bit."""
    torch = torch_cuda
    K = synth.K_matrix(W, H)
    kinv = np.linalg.inv(K.astype(np.float64)).astype(np.float32)
    plan = vdist.ShardPlan(KW["numBuckets"], world)
    shards = [vdist.HipShard(vh.default_params(**KW), W, H, sem, plan, r, W * H // 4, batch=batch,
                             batched_calls=(calls == "batched"), sensor_k_inv=kinv) for r in range(world)]
    full = oracle.OracleTable(oracle.default_params(**KW), W, H, sem)
    for step in range(0, 4, batch):
        frames = []
        for b in range(batch):
            cams = []
            for pose, verts in cameras(world, step + b):
                d16 = np.round(verts[..., 2] * 5000.0).clip(0, 65535).astype(np.uint16)
                d16[::9, ::7] = 0                                       # sensor holes
                v = oracle.preprocess(d16, kinv)[0]
                cams.append((pose, v, d16))
            frames.append(cams)
        vdist.loopback_step(shards, [[frames[b][r][0] for b in range(batch)] for r in range(world)],
                            [[torch.from_numpy(frames[b][r][1]).cuda() for b in range(batch)] for r in range(world)],
                            [[torch.from_numpy(frames[b][r][2]).cuda() for b in range(batch)] for r in range(world)])
        for cams in frames:
            vdist.reference_multi_camera_frame(full, [c[0] for c in cams], [c[1] for c in cams])
    total = 0
    for r, sh in enumerate(shards):
        sh.table.synchronize()
        total += check_shard_against_full(sh.table, full, *plan.bucket_range(r), 5)
        assert sh.table.counters()["bin_overflow"] == 0
        assert sh.packet_floats == 36 + W * H // 2
    assert total == len(full.allocated())
    if sem == 1:
        assert total > 100
    for sh in shards:
        sh.table.close()


@pytest.mark.parametrize("world,batch,mode", [(2, 3, 1), (4, 2, 0), (2, 5, 2)])
def test_one_key_bin_per_owner_and_batch(oracle, vh, torch_cuda, world, batch, mode):
    """VH_BIN_PER_BATCH: the keys of all frames of a batch travel in ONE bin per owner, each record with its frame index
    where a per-frame bin's record has the camera id; the launch of frame b claims the records of frame b.  Same tables as
    with per-frame bins = the one oracle table, with pipelined launches (1), two launches per frame (0) and the last frame
    pending across calls (2), with an allocation band (several keys per pixel: the sample bits of the rank)."""
    torch = torch_cuda
    K = synth.K_matrix(W, H)
    kinv = np.linalg.inv(K.astype(np.float64)).astype(np.float32)
    plan = vdist.ShardPlan(KW["numBuckets"], world)
    kw = dict(numBuckets=KW["numBuckets"], numVoxelBlocks=1 << 14)       # (per-shard heaps that the band never exhausts)
    shards = [vdist.HipShard(vh.default_params(**kw), W, H, 1, plan, r, batch * W * H // 8, batch=batch, sensor_k_inv=kinv,
                             per_batch_bins=True, sets=2) for r in range(world)]
    full = oracle.OracleTable(oracle.default_params(**kw), W, H, 1)
    full.set_alloc_band(0.1)
    for sh in shards:
        sh.table.set_option("pipeline_shards", mode)
        sh.table.set_alloc_band(0.1)
        assert sh.bins_send.shape == (world, 1, batch * W * H // 8, 4)
    fills = []
    for step in range(0, 2 * batch, batch):
        for sh in shards:                                # (mode 2: the previous batch's last packets stay where they are)
            sh.use_set((step // batch) % 2)
        frames = []
        for b in range(batch):
            cams = []
            for pose, verts in cameras(world, step + b):
                d16 = np.round(verts[..., 2] * 5000.0).clip(0, 65535).astype(np.uint16)
                cams.append((pose, oracle.preprocess(d16, kinv)[0], d16))
            frames.append(cams)
        vdist.loopback_step(shards, [[frames[b][r][0] for b in range(batch)] for r in range(world)],
                            [[None] * batch for _ in range(world)],
                            [[torch.from_numpy(frames[b][r][2]).cuda() for b in range(batch)] for r in range(world)])
        fills.append(int(shards[0].bins_recv[:, 0, 0, 0].sum().item()))
        for cams in frames:
            vdist.reference_multi_camera_frame(full, [c[0] for c in cams], [c[1] for c in cams])
    total = 0
    for r, sh in enumerate(shards):
        sh.table.synchronize()
        total += check_shard_against_full(sh.table, full, *plan.bucket_range(r), 5)
        assert sh.table.counters()["bin_overflow"] == 0 and sh.table.counters()["heap_exhausted"] == 0
    assert total == len(full.allocated()) > 300
    assert min(fills) > 100 * batch                      # (the bins really carried the whole batch)
    for sh in shards:
        sh.table.close()


def _hip_gloo_worker(rank, world, port, q, sensor):
    """One rank of a two-process run on ONE GPU: real HIP shards, real processes, gloo with host staging
    for the exchange (RCCL refuses two ranks on one device).  Checks its shard against the oracle."""
    try:
        import os
        import sys
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import torch
        import torch.distributed as dist
        import oracle as O
        import voxelhashing_demo_amd as V
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        kinv = np.linalg.inv(synth.K_matrix(W, H).astype(np.float64)).astype(np.float32)
        plan = vdist.ShardPlan(KW["numBuckets"], world)
        batch = 2
        table_stream, front = torch.cuda.Stream(), torch.cuda.Stream()
        shard = vdist.HipShard(V.default_params(**KW), W, H, 1, plan, rank, W * H // 4, batch=batch, stream=table_stream,
                               sets=2, sensor_k_inv=kinv if sensor else None)
        transport = vdist.TorchDistTransport()
        pipe = vdist.ShardedPipeline(shard, transport, table_stream, front)
        full = O.OracleTable(O.default_params(**KW), W, H, 1)
        for step in range(0, 6, batch):
            frames = []
            for b in range(batch):
                cams = []
                for pose, verts in cameras(world, step + b):
                    d16 = None
                    if sensor:
                        d16 = np.round(verts[..., 2] * 5000.0).clip(0, 65535).astype(np.uint16)
                        verts = O.preprocess(d16, kinv)[0]
                    cams.append((pose, verts, d16))
                frames.append(cams)
            pipe.feed([f[rank][0] for f in frames], [torch.from_numpy(f[rank][1]).cuda() for f in frames],
                      [torch.from_numpy(f[rank][2]).cuda() for f in frames] if sensor else None)
            for cams in frames:
                vdist.reference_multi_camera_frame(full, [c[0] for c in cams], [c[1] for c in cams])
        pipe.flush()
        n = check_shard_against_full(shard.table, full, *plan.bucket_range(rank), 5)
        # raycast over the two shards: this rank's last view
        view = vdist.HipViewTable(V.default_params(**KW), W, H, 1, world, 4096)
        pose = frames[-1][rank][0]
        depth, lost = vdist.sharded_raycast(shard, view, transport, pose, 4096)
        torch.cuda.synchronize()
        assert lost == 0 and np.array_equal(depth.cpu().numpy().view(np.uint32), full.raycast(pose).view(np.uint32))
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok", n, len(full.allocated())))
    except Exception as e:
        import traceback
        q.put((rank, "fail", traceback.format_exc(), str(e)))


@pytest.mark.parametrize("sensor", [True])      # (float packets between two processes: tests/test_gpu_dist_rccl.py [2-1-False-False])
def test_two_processes_on_one_gpu(oracle, vh, torch_cuda, sensor):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_hip_gloo_worker, args=(r, 2, port, q, sensor)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    for r in results:
        assert r[1] == "ok", r[2]
    assert sum(r[2] for r in results) == results[0][3] > 100


def test_one_launch_multi_camera_frames(oracle, vh, torch_cuda):
    """vh_apply_frames_batch runs a batch of B multi-camera frames as B + 1 launches (frame_multi_pipelined_kernel: the
    commit + TSDF update of frame b ride in the launch of frame b + 1; option "pipeline_shards", on by default), as B
    (option 2: the last frame's half rides in the first launch of the next batch) or as 2 B (option off).  Same frames through both on two shards: the same tables, equal to the slices of ONE oracle table --
    with batches of 1, 2 and 5, new blocks in every frame (the cameras move), a collection in between and single-camera
    pipelined frames on the same contexts before and after (the two pipelines share their buffers and counter sets)."""
    torch = torch_cuda
    world = 2
    plan = vdist.ShardPlan(KW["numBuckets"], world)
    full = oracle.OracleTable(oracle.default_params(**KW), W, H, 1)
    runs = {}
    keep = []            # (pipeline_shards 2: the last frame's packets stay valid until the next call or flush)
    for mode in (1, 0, 2):
        shards = [vdist.HipShard(vh.default_params(**KW), W, H, 1, plan, r, W * H // 4, batch=5) for r in range(world)]
        for sh in shards:
            sh.table.set_option("pipeline_shards", mode)
        step = 0
        for batch in (1, 2, 5, 1, 5):
            for sh in shards:
                sh.batch = batch
            frames = [cameras(world, step + b) for b in range(batch)]
            # (HipShard's buffers are sized for batch 5: smaller batches use the dense prefix)
            for r, sh in enumerate(shards):
                sh.table.set_pose(frames[0][r][0])
            sub = [vdist.HipShard.__new__(vdist.HipShard) for _ in shards]
            for s2, sh in zip(sub, shards):
                s2.__dict__.update(sh.__dict__)
                s2.batch = batch
                s2.bins_send, s2.bins_recv = sh.bins_send[:, :batch].contiguous(), sh.bins_recv[:, :batch].contiguous()
                s2.packet, s2.packets = sh.packet[:batch].contiguous(), sh.packets[:, :batch].contiguous()
            keep.append(sub)
            vdist.loopback_step(sub, [[frames[b][r][0] for b in range(batch)] for r in range(world)],
                                [[torch.from_numpy(frames[b][r][1]).cuda() for b in range(batch)] for r in range(world)])
            if mode == 1:
                for cams in frames:
                    vdist.reference_multi_camera_frame(full, [c[0] for c in cams], [c[1] for c in cams])
            step += batch
            if step == 3:
                for sh in shards:
                    sh.table.garbage_collect(0.05)
                if mode == 1:
                    full.garbage_collect(0.05)
        for sh in shards:
            sh.table.synchronize()
        runs[mode] = shards
    total = 0
    for r in range(world):
        lo, hi = plan.bucket_range(r)
        total += check_shard_against_full(runs[1][r].table, full, lo, hi, 5)
        for other in (0, 2):
            check_shard_against_full(runs[other][r].table, full, lo, hi, 5)
            assert runs[1][r].table.counters()["occupied"] == runs[other][r].table.counters()["occupied"]
            assert entries_as_set_(runs[1][r].table.compact()) == entries_as_set_(runs[other][r].table.compact())
    assert total == len(full.allocated()) > 100
    for m in runs:
        for sh in runs[m]:
            sh.table.close()


def entries_as_set_(entries):
    return set(map(tuple, np.asarray(entries["pos"]).reshape(-1, 3).tolist()))


def test_one_launch_multi_camera_frames_across_the_epoch_wrap(oracle, vh, torch_cuda):
    """The claim words carry a 9-bit lock epoch; at the wrap they are cleared -- which the frame whose deferred half is
    still pending must not see: 1 040 multi-camera frames (one shard, batches of 8) straddle the wraps at frames 511 and 1 022."""
    torch = torch_cuda
    kw = dict(numBuckets=1 << 10, numVoxelBlocks=4096)            # (a heap that never runs dry: 1 024 blocks did)
    w, h = 64, 48
    plan = vdist.ShardPlan(kw["numBuckets"], 1)
    prims = synth.room_primitives()
    poses = synth.camera_loop(40)
    verts = [synth.render_room_verts(p, w, h, prims).numpy() for p in poses]
    dv = [torch.from_numpy(v).cuda() for v in verts]
    sh = vdist.HipShard(vh.default_params(**kw), w, h, 1, plan, 0, w * h // 2, batch=8)
    full = oracle.OracleTable(oracle.default_params(**kw), w, h, 1)
    for step in range(130):
        ks = [(8 * step + b) % 40 for b in range(8)]
        vdist.loopback_step([sh], [[poses[k] for k in ks]], [[dv[k] for k in ks]])
        for k in ks:
            full.integrate(poses[k], verts[k])
    sh.table.synchronize()
    assert sh.table.counters()["epoch"] == 1040 and sh.table.counters()["heap_exhausted"] == 0
    assert check_shard_against_full(sh.table, full, 0, kw["numBuckets"], 5) > 20
    sh.table.close()


def test_single_camera_pipelined_frames_between_deferred_multi_camera_halves(oracle, vh, torch_cuda):
    """a context that holds a multi-camera frame's deferred half (pipeline_shards 2, what vh_dist's shard is set
    to) is handed pipelined single-camera frames (vh_integrate, option "pipeline"): the deferred half must be served first --
    the two pipelines share buffer parity and counter sets -- and the other way round at the next multi-camera batch."""
    torch = torch_cuda
    plan = vdist.ShardPlan(KW["numBuckets"], 1)
    sh = vdist.HipShard(vh.default_params(**KW), W, H, 1, plan, 0, W * H // 4, batch=2)
    sh.table.set_option("pipeline_shards", 2)
    sh.table.set_option("pipeline", 1)
    full = oracle.OracleTable(oracle.default_params(**KW), W, H, 1)
    keep, step = [], 0
    for round_ in range(3):
        frames = [cameras(1, step + b) for b in range(2)]
        dv = [[torch.from_numpy(frames[b][0][1]).cuda() for b in range(2)]]
        keep.append(dv)
        vdist.loopback_step([sh], [[frames[b][0][0] for b in range(2)]], dv)           # leaves the last frame's half pending
        for cams in frames:
            full.integrate(cams[0][0], cams[0][1])
        step += 2
        for _ in range(1 + round_):                                                    # 1, 2, 3 pipelined single-camera frames
            (pose, verts), = cameras(1, step)
            d = torch.from_numpy(verts).cuda()
            keep.append(d)
            sh.table.integrate(pose, d)
            full.integrate(pose, verts)
            step += 1
    sh.table.synchronize()
    assert check_shard_against_full(sh.table, full, 0, KW["numBuckets"], 5) > 100
    assert sh.table.counters()["occupied"] == len(full.compact())
    sh.table.close()
    full.close()
