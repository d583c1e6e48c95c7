"""Raycast over bucket-range shards on the GPU (vh_export_views / vh_import_view / vh_raycast):
R shard contexts and R view contexts on one device with the in-process exchange, and the RCCL
transport with one rank.  The depth images must equal the ORACLE's raycast of ONE unsharded
table bit for bit."""
import numpy as np
import pytest

from test_sharding_cpu import _free_port
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


def build(oracle, vh, torch, world, sem, steps=3):
    plan = vdist.ShardPlan(KW["numBuckets"], world)
    shards = [vdist.HipShard(vh.default_params(**KW), W, H, sem, plan, r, W * H // 4) for r in range(world)]
    full = oracle.OracleTable(oracle.default_params(**KW), W, H, sem)
    for step in range(steps):
        cams = cameras(world, step)
        vdist.loopback_step(shards, [[c[0]] for c in cams], [[torch.from_numpy(c[1]).cuda()] for c in cams])
        vdist.reference_multi_camera_frame(full, [c[0] for c in cams], [c[1] for c in cams])
    for sh in shards:
        sh.table.synchronize()
    return plan, shards, full


@pytest.mark.parametrize("world", [1, 2, 4])
@pytest.mark.parametrize("sem", [0, 1])
def test_sharded_raycast_equals_oracle_raycast_of_one_table(oracle, vh, torch_cuda, world, sem):
    torch = torch_cuda
    plan, shards, full = build(oracle, vh, torch, world, sem)
    views = [vdist.HipViewTable(vh.default_params(**KW), W, H, sem, world, 2048) for _ in range(world)]
    hits = 0
    for round_ in range(2):                      # the second round re-imports into used view tables
        poses = [c[0] for c in cameras(world, 2 - round_)]
        depths = vdist.loopback_raycast(shards, views, poses, capacity=2048)
        for r in range(world):
            ref = full.raycast(poses[r])
            assert np.array_equal(depths[r].view(np.uint32), ref.view(np.uint32)), (round_, r)
            hits += int((ref > 0).sum())
            # the block silhouettes of a view table: every cube an in-image ray meets inside the depth range
            # was selected for the view, so they equal the silhouettes of the one unsharded table
            front, back = torch.empty((H, W), device="cuda"), torch.empty((H, W), device="cuda")
            views[r].table.render_blocks(poses[r], front, back, 0.1, 5.0)
            torch.cuda.synchronize()
            of, ob = full.render_blocks(poses[r], 0.1, 5.0)
            assert np.array_equal(front.cpu().numpy().view(np.uint32), of.view(np.uint32)), (round_, r)
            assert np.array_equal(back.cpu().numpy().view(np.uint32), ob.view(np.uint32)), (round_, r)
        for v in views:
            assert v.table.counters()["bin_overflow"] == 0
    if sem == 1:
        assert hits > 10000 * world
    for x in shards + views:
        x.table.close()


@pytest.mark.parametrize("world", [1, 2, 4])
def test_fixed_slot_round_equals_oracle_raycast(oracle, vh, torch_cuda, world):
    """The sync-free form (vh_export_views_fixed / vh_import_views: device poses, the view frustums
    prepared on the device, fixed record slots, counts read on the device) renders the same bits, and
    selects exactly the sets the host-pose form selects."""
    torch = torch_cuda
    plan, shards, full = build(oracle, vh, torch, world, 1)
    views = [vdist.HipViewTable(vh.default_params(**KW), W, H, 1, world, 2048) for _ in range(world)]
    for round_ in range(2):
        poses = [c[0] for c in cameras(world, 2 - round_)]
        depths = vdist.loopback_raycast_fixed(shards, views, poses, capacity=2048)
        for r in range(world):
            ref = full.raycast(poses[r])
            assert np.array_equal(depths[r].view(np.uint32), ref.view(np.uint32)), (round_, r)
        for v in views:
            assert v.table.counters()["bin_overflow"] == 0
    # same selection as the host-pose export
    d_poses = torch.from_numpy(np.asarray(poses, np.float32).reshape(world, 16)).cuda()
    for sh in shards:
        rec_a, cnt_a = sh.export_views(poses, 2048)
        cnt_a = cnt_a.cpu().numpy().copy()
        keys_a, first = [], 0
        ra = rec_a.cpu().numpy()
        for v in range(world):
            keys_a.append({tuple(int(c) for c in g[:12].view(np.int32)) for g in ra[first:first + cnt_a[v]]})
            first += cnt_a[v]
        send = torch.zeros((world * 2048, 4112), dtype=torch.uint8, device="cuda")
        cnt = torch.zeros((world,), dtype=torch.int32, device="cuda")
        sh.table.export_views_fixed(d_poses, world, send, 2048, cnt)
        torch.cuda.synchronize()
        cnt_b, rb = cnt.cpu().numpy(), send.cpu().numpy().reshape(world, 2048, 4112)
        assert np.array_equal(cnt_a, cnt_b)
        for v in range(world):
            assert {tuple(int(c) for c in g[:12].view(np.int32)) for g in rb[v, :cnt_b[v]]} == keys_a[v]
    # a capacity that is too small is reported, not hidden
    small = vdist.HipViewTable(vh.default_params(**KW), W, H, 1, world, 16)
    send = torch.zeros((world * 16, 4112), dtype=torch.uint8, device="cuda")
    cnt = torch.zeros((world,), dtype=torch.int32, device="cuda")
    shards[0].table.export_views_fixed(d_poses, world, send, 16, cnt)
    recv = send[:16].repeat(world, 1).contiguous()
    cin = cnt[:1].repeat(world).contiguous()
    small.table.import_views(recv, world, 16, cin)
    torch.cuda.synchronize()
    # (every source range holds the same 16 records here, so duplicates add bucket-full losses on top)
    assert int(cnt[0]) > 16 and small.table.counters()["bin_overflow"] >= world * (int(cnt[0]) - 16)
    for x in shards + views + [small]:
        x.table.close()


def test_export_equals_oracle_export(oracle, vh, torch_cuda):
    """Same selected set, same voxel bytes (the order inside a view is free)."""
    torch = torch_cuda
    world = 2
    plan, shards, full = build(oracle, vh, torch, world, 1)
    oshards = []
    for r in range(world):
        lo, hi = plan.bucket_range(r)
        o = oracle.OracleTable(oracle.default_params(**KW), W, H, 1, bucket_range=(lo, hi))
        oshards.append(o)
    poses = [c[0] for c in cameras(3, 1)]        # three views, two shards
    for r, sh in enumerate(shards):
        records, counts = sh.export_views(poses, 2048)
        torch.cuda.synchronize()
        counts = counts.cpu().numpy()
        rec = records.cpu().numpy()
        first = 0
        for v, pose in enumerate(poses):
            got = rec[first:first + counts[v]]
            first += counts[v]
            got_map = {tuple(int(c) for c in g[:12].view(np.int32)): g[16:].tobytes() for g in got}
            assert len(got_map) == counts[v]
            # oracle selection over the same shard content (downloaded from the GPU shard)
            table = sh.table.hash_table()
            blocks = sh.table.sdf_blocks()
            want = {}
            f = oshards[r].view_frustum(pose)
            for e in table[table["ptr"] != -1]:
                if oshards[r].view_holds_block(f, e["pos"]):
                    want[tuple(int(c) for c in e["pos"])] = blocks[int(e["ptr"]):int(e["ptr"]) + 512].tobytes()
            assert got_map == want
            assert len(want) > 20


def test_capacity_overflow_is_reported_and_bounded(oracle, vh, torch_cuda):
    torch = torch_cuda
    plan, shards, full = build(oracle, vh, torch, 1, 1, steps=2)
    poses = [cameras(1, 1)[0][0]]
    records, counts = shards[0].export_views(poses, 7)
    torch.cuda.synchronize()
    assert int(counts[0]) > 7
    keys = records[:7, :12].cpu().numpy().view(np.int32)
    allocated = {tuple(k) for k in full.allocated()["pos"].tolist()}
    assert all(tuple(k) in allocated for k in keys.tolist())


def test_more_views_than_one_launch_takes(oracle, vh, torch_cuda):
    """20 views: the select walk runs twice (16 views per launch)."""
    torch = torch_cuda
    plan, shards, full = build(oracle, vh, torch, 1, 1, steps=2)
    poses = [synth.camera_loop(60)[3 * i] for i in range(20)]
    view = vdist.HipViewTable(vh.default_params(**KW), W, H, 1, 1, 4096)
    records, counts = shards[0].export_views(poses, 4096)
    torch.cuda.synchronize()
    counts = counts.cpu().tolist()
    first = 0
    for v in (0, 15, 16, 19):
        first = sum(counts[:v])
        view.recv[:counts[v]].copy_(records[first:first + counts[v]])
        depth = view.render(counts[v], poses[v]).cpu().numpy()
        assert np.array_equal(depth, full.raycast(poses[v])), v


def test_empty_import_and_argument_checks(oracle, vh, torch_cuda):
    torch = torch_cuda
    view = vdist.HipViewTable(vh.default_params(**KW), W, H, 1, 1, 16)
    depth = view.render(0, synth.camera_loop(60)[0]).cpu().numpy()
    assert not depth.any()
    used = vh.SDFHashtable(vh.default_params(**KW), W, H, 1)
    pose, verts = cameras(1, 0)[0]
    used.integrate(pose, torch.from_numpy(verts).cuda())
    with pytest.raises(RuntimeError):
        used.import_view(view.recv, 0)           # a table that integrated frames is not a view table
    plan = vdist.ShardPlan(KW["numBuckets"], 2)
    shard = vh.SDFHashtable(vh.default_params(**KW), W, H, 1, bucket_range=plan.bucket_range(1))
    with pytest.raises(RuntimeError):
        shard.import_view(view.recv, 0)          # nor is a shard


def test_sharded_raycast_over_nccl(oracle, vh, torch_cuda):
    import os

    import torch.distributed as dist
    torch = torch_cuda
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(_free_port())        # a fixed port may still be held by an earlier run
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        plan, shards, full = build(oracle, vh, torch, 1, 1)
        view = vdist.HipViewTable(vh.default_params(**KW), W, H, 1, 1, 4096)
        transport = vdist.TorchDistTransport()
        for step in (2, 0):
            pose = cameras(1, step)[0][0]
            depth, lost = vdist.sharded_raycast(shards[0], view, transport, pose, 4096)
            torch.cuda.synchronize()
            ref = full.raycast(pose)
            assert lost == 0 and np.array_equal(depth.cpu().numpy(), ref) and (ref > 0).sum() > 10000
    finally:
        dist.destroy_process_group()
