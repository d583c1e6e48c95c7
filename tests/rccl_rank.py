"""One rank of tests/test_gpu_dist_rccl.py: the native multi-GPU host (include/voxelhash_dist.h) on its DEFAULT transport,
RCCL, with R > 1 ranks -- one process per rank, all on cuda:0.

RCCL refuses two ranks of one host on one device ("Duplicate GPU detected"), but only ranks of one HOST: with a different
NCCL_HOSTID per rank the ranks look like R single-GPU nodes and talk through RCCL's socket transport over the loop-back
interface.  Slow, and nothing a deployment would do -- but every ncclCommInitRank / ncclSend / ncclRecv / ncclAllGather
the library issues is the real one, which no one-GPU box could run otherwise.  The test sets the environment before this
process starts; the control plane (the unique id, the verdicts) is torch.distributed over gloo.

usage: rccl_rank.py RANK WORLD PORT BATCH SENSOR(0|1) [raycast_auto] [size=WxH] [buckets=LOG2] [blocks=LOG2] [exchanges=N]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world, port, batch, sensor = (int(a) for a in sys.argv[1:6])
    auto = "raycast_auto" in sys.argv[6:]
    opt = dict(a.split("=") for a in sys.argv[6:] if "=" in a)
    import numpy as np
    import torch
    import torch.distributed as dist

    import oracle
    import voxelhashing_demo_amd as vh
    from test_sharding_cpu import check_shard_against_full
    from voxelhashing_demo_amd import dist as vdist
    from voxelhashing_demo_amd import synth

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world, init_method=f"tcp://127.0.0.1:{port}")
    W, H = (int(x) for x in opt.get("size", "320x240").split("x"))
    kw = dict(numBuckets=1 << int(opt.get("buckets", 14)), numVoxelBlocks=1 << int(opt.get("blocks", 13)))
    steps = int(opt.get("exchanges", 4)) * batch
    prims = synth.room_primitives()
    kinv = np.linalg.inv(synth.K_matrix(W, H).astype(np.float64)).astype(np.float32)
    frames = []                                       # frames[s][r] = (pose, verts as the oracle sees them, what rank r feeds)
    for s in range(steps):
        cams = []
        for r in range(world):
            pose = synth.camera_loop(40, phase=vdist.camera_phase(r, world))[(3 * s) % 40]
            v = synth.render_room_verts(pose, W, H, prims).numpy()
            if sensor:
                d16 = np.round(v[..., 2] * 5000.0).clip(0, 65535).astype(np.uint16)
                v = oracle.preprocess(d16, kinv)[0]
                cams.append((pose, v, d16))
            else:
                cams.append((pose, v, v))
        frames.append(cams)
    uid = vdist.unique_id(rank, vdist.torch_broadcast_bytes())
    nd = vdist.NativeDist(vh.default_params(**kw), W, H, 1, rank, world, batch, uid, sensor_k_inv=kinv if sensor else None)
    assert nd.transport == "rccl" and nd.comm_info() == (rank, world), (nd.transport, nd.comm_info())
    if os.environ.get("VOXELHASH_DIST_FUSED") == "2":     # the key generation inside the frame launches: only the reference's walk carries it
        nd.table.set_option("flatten_variant", 3)
    mine = [torch.from_numpy(np.ascontiguousarray(frames[s][rank][2])).cuda() for s in range(steps)]
    torch.cuda.synchronize()
    for s0 in range(0, steps, batch):
        nd.step([frames[s0 + b][rank][0] for b in range(batch)], mine[s0:s0 + batch])
    nd.flush()
    # the oracle: ONE unsharded table, every multi-camera frame in order
    full = oracle.OracleTable(oracle.default_params(**kw), W, H, 1)
    for cams in frames:
        vdist.reference_multi_camera_frame(full, [c[0] for c in cams], [c[1] for c in cams])
    lo, hi = vdist.ShardPlan(kw["numBuckets"], world).bucket_range(rank)
    n = check_shard_against_full(nd.table, full, lo, hi, 5)
    c = nd.table.counters()
    assert c["bin_overflow"] == 0 and c["epoch"] == steps, c
    counts = [None] * world
    dist.all_gather_object(counts, n)
    assert sum(counts) == len(full.allocated()) > 100, (counts, len(full.allocated()))
    # a raycast round over the shards: every rank renders ITS camera's view of the whole table
    pose = frames[-1][rank][0]
    out = torch.empty((H, W), dtype=torch.float32, device="cuda")
    want = full.raycast(pose)
    if auto:
        normals = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
        cap = nd.raycast_auto(pose, out, normals)
        wd, wn = full.raycast(pose, normals=True)
        assert np.array_equal(normals.cpu().numpy().view(np.uint32), wn.view(np.uint32)), "normals"
        assert cap >= 1
    else:
        lost = torch.zeros(1, dtype=torch.int32, device="cuda")
        for _ in range(2):                              # (back to back: the view table and the slot buffers are reused)
            nd.raycast(pose, out, 2048, lost=lost)
        nd.flush()
        torch.cuda.synchronize()
        assert int(lost.item()) == 0
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32)), "raycast"
    assert (want > 0).mean() > 0.3
    # more exchanges behind the raycast round (the buffer-set rotation restarts)
    for s0 in range(0, 2 * batch, batch):
        nd.step([frames[s0 + b][rank][0] for b in range(batch)], mine[s0:s0 + batch])
    nd.flush()
    for cams in frames[:2 * batch]:
        vdist.reference_multi_camera_frame(full, [c[0] for c in cams], [c[1] for c in cams])
    check_shard_against_full(nd.table, full, lo, hi, 5)
    dist.barrier()
    nd.close()
    full.close()
    dist.destroy_process_group()
    print(f"rank {rank}: ok, {n} blocks of {sum(counts)}", flush=True)


if __name__ == "__main__":
    main()
