"""The N>1 path on CPU: bucket-range sharding with the oracle as the compute backend,
(a) all ranks played in one process, (b) two real processes over torch.distributed "gloo".
A sharded step must equal the multi-camera frame on ONE unsharded table."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import blocks_by_pos
from oracle_shards import OracleShard, OracleViewTable
from voxelhashing_demo_amd import dist as vdist
from voxelhashing_demo_amd import synth

W, H = 160, 120
KW = dict(numBuckets=1 << 12, numVoxelBlocks=8192)   # enough blocks: heap exhaustion is not under test


def camera_inputs(world, step):
    """Deterministic per-camera pose + vertex map (every rank can rebuild all of them)."""
    prims = synth.room_primitives()
    out = []
    for r in range(world):
        pose = synth.camera_loop(40, phase=vdist.camera_phase(r, world))[(3 * step) % 40]
        out.append((pose, synth.render_room_verts(pose, W, H, prims).numpy()))
    return out


def check_shard_against_full(shard_table, full_table, lo, hi, bs):
    full = full_table.hash_table()[lo * bs:hi * bs]
    mine = shard_table.hash_table()
    assert np.array_equal(full["pos"], mine["pos"])
    assert np.array_equal(full["ptr"] != -1, mine["ptr"] != -1)
    fb = blocks_by_pos(full[full["ptr"] != -1], full_table.sdf_blocks())
    mb = blocks_by_pos(mine[mine["ptr"] != -1], shard_table.sdf_blocks())
    assert fb.keys() == mb.keys()
    for pos in fb:
        assert np.array_equal(fb[pos].view(np.uint32), mb[pos].view(np.uint32)), pos
    return len(fb)


def test_shard_plan():
    p = vdist.ShardPlan(1 << 20, 8)
    assert [p.bucket_range(r) for r in (0, 7)] == [(0, 1 << 17), (7 << 17, 1 << 20)]
    p = vdist.ShardPlan(10, 4)          # ragged: 3,3,3,1
    assert [p.bucket_range(r) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert [p.owner(h) for h in (0, 2, 3, 8, 9)] == [0, 0, 1, 2, 3]
    with pytest.raises(ValueError):
        vdist.ShardPlan(2, 4)


@pytest.mark.parametrize("world,batch", [(1, 1), (2, 1), (3, 1), (2, 2)])
@pytest.mark.parametrize("sem", [0, 1])
def test_loopback_shards_equal_one_table(oracle, world, batch, sem):
    """batch > 1: several frames per camera travel in one exchange and are applied in order;
    the result is the same as one multi-camera frame per exchange."""
    plan = vdist.ShardPlan(KW["numBuckets"], world)
    shards = [OracleShard(oracle, oracle.default_params(**KW), W, H, sem, plan, r, W * H + 1, batch=batch)
              for r in range(world)]
    full = oracle.OracleTable(oracle.default_params(**KW), W, H, sem)
    total = 0
    for step in range(0, 4, batch):
        frames = [camera_inputs(world, step + b) for b in range(batch)]          # frames[b][r]
        vdist.loopback_step(shards, [[frames[b][r][0] for b in range(batch)] for r in range(world)],
                            [[frames[b][r][1] for b in range(batch)] for r in range(world)])
        for cams in frames:
            vdist.reference_multi_camera_frame(full, [c[0] for c in cams], [c[1] for c in cams])
    for r, sh in enumerate(shards):
        lo, hi = plan.bucket_range(r)
        total += check_shard_against_full(sh.table, full, lo, hi, KW.get("bucketSize", 5))
    assert total == len(full.allocated()) and (total > 20 or sem == 0)


def test_one_camera_multi_frame_is_the_reference_integrate(oracle):
    """R = 1: the multi-camera frame is SDF_Hashtable::integrate."""
    a = oracle.OracleTable(oracle.default_params(**KW), W, H, 1)
    b = oracle.OracleTable(oracle.default_params(**KW), W, H, 1)
    for step in range(3):
        (pose, verts), = camera_inputs(1, step)
        a.integrate(pose, verts)
        vdist.reference_multi_camera_frame(b, [pose], [verts])
    check_shard_against_full(b, a, 0, KW["numBuckets"], 5)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _gloo_worker(rank, world, port, sem, q, batch=2):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        import torch.distributed as dist
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import oracle as O
        dist.init_process_group("gloo", rank=rank, world_size=world)
        plan = vdist.ShardPlan(KW["numBuckets"], world)
        shard = OracleShard(O, O.default_params(**KW), W, H, sem, plan, rank, W * H + 1, batch=batch)
        full = O.OracleTable(O.default_params(**KW), W, H, sem)
        transport = vdist.TorchDistTransport()
        for step in range(0, 4, batch):
            frames = [camera_inputs(world, step + b) for b in range(batch)]
            vdist.sharded_step(shard, transport, [f[rank][0] for f in frames], [f[rank][1] for f in frames])
            for cams in frames:
                vdist.reference_multi_camera_frame(full, [c[0] for c in cams], [c[1] for c in cams])
        lo, hi = plan.bucket_range(rank)
        n = check_shard_against_full(shard.table, full, lo, hi, 5)
        # raycast over the shards: this rank's own view, blocks gathered from both ranks
        view = OracleViewTable(O, O.default_params(**KW), W, H, sem, world, 2048)
        pose = frames[-1][rank][0]
        depth, lost = vdist.sharded_raycast(shard, view, transport, pose, 2048)
        ref = full.raycast(pose)
        assert lost == 0 and np.array_equal(depth.numpy(), ref) and (ref > 0).sum() > 1000
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok", n, len(full.allocated())))
    except Exception as e:   # surface the failure to the parent
        import traceback
        q.put((rank, "fail", traceback.format_exc(), str(e)))


@pytest.mark.parametrize("sem", [1])
def test_two_processes_over_gloo(oracle, sem):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, sem, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for r in results:
        assert r[1] == "ok", r[2]
    assert sum(r[2] for r in results) == results[0][3] > 20
