"""Block deletion / garbage collection on the GPU (vh_delete_blocks, vh_garbage_collect) against
the oracle: same allocated positions in the same slots after every step, same voxel bits, same
heap counter; freed blocks are zero and sit on the heap."""
import numpy as np
import pytest

from conftest import blocks_by_pos
from voxelhashing_demo_amd import dist as vdist
from voxelhashing_demo_amd import synth

pytestmark = pytest.mark.gpu

W, H = 320, 240
VARIANTS = {"fused-ballot-walk": (1, 3), "four-kernel-reference-walk": (0, 3), "fused-indexed-walk": (1, 4)}


def frames(n, step=5):
    prims = synth.room_primitives()
    poses = synth.camera_loop(60)
    return [(poses[(step * i) % 60], synth.render_room_verts(poses[(step * i) % 60], W, H, prims).numpy())
            for i in range(n)]


def pair(oracle, vh, variant, band=0.0, **kw):
    ot = oracle.OracleTable(oracle.default_params(**kw), W, H, 1)
    gt = vh.SDFHashtable(vh.default_params(**kw), W, H, 1)
    fused, walk = VARIANTS[variant]
    gt.set_option("fused_frame", fused)
    gt.set_option("flatten_variant", walk)
    if band:
        ot.set_alloc_band(band)
        gt.set_alloc_band(band)
    return ot, gt


def same_state(ot, gt, bs=5):
    gt.synchronize()
    a, b = ot.hash_table(), gt.hash_table()
    assert np.array_equal(a["pos"], b["pos"])
    assert np.array_equal(a["ptr"] != -1, b["ptr"] != -1)
    assert np.array_equal(a["offset"], b["offset"])
    oa = blocks_by_pos(a[a["ptr"] != -1], ot.sdf_blocks())
    ga = blocks_by_pos(b[b["ptr"] != -1], gt.sdf_blocks())
    assert oa.keys() == ga.keys()
    for k in oa:
        assert np.array_equal(oa[k].view(np.uint32), ga[k].view(np.uint32)), k
    c = gt.counters()
    assert c["heap_counter"] == ot.heap_counter()
    # GPU heap: free ids + referenced ids partition the pool, free blocks are zero
    live = b["ptr"][b["ptr"] != -1] // 512
    free = gt.heap()[:c["heap_counter"] + 1]
    n = gt.params.numVoxelBlocks
    assert sorted(live.tolist() + free.tolist()) == list(range(n))
    vol = gt.sdf_blocks().view(np.uint32).reshape(n, 1024)
    assert not vol[free].any()
    return len(oa)


@pytest.mark.parametrize("variant", list(VARIANTS))
def test_delete_blocks(oracle, vh, torch_cuda, variant):
    torch = torch_cuda
    ot, gt = pair(oracle, vh, variant, numBuckets=256, bucketSize=5, numVoxelBlocks=1024)
    fr = frames(6)
    for pose, verts in fr:
        ot.integrate(pose, verts)
        gt.integrate(pose, torch.from_numpy(verts).cuda())
    n0 = same_state(ot, gt)
    keys = sorted(tuple(k) for k in ot.allocated()["pos"].tolist())
    victims = keys[::3] + [(999, 999, 999)] + keys[:2]            # an absent key and two listed twice
    assert ot.delete_blocks(victims) == len(keys[::3]) + 1        # keys[0] is in both lists, keys[1] only in the second
    k4 = np.zeros((len(victims), 4), np.int32)
    k4[:, :3] = victims
    gt.delete_blocks(torch.from_numpy(k4).cuda())
    n1 = same_state(ot, gt)
    assert n1 == n0 - len(keys[::3]) - 1
    c = gt.counters()
    assert c["last_freed"] == n0 - n1 == c["freed_total"] and c["occupied"] == 0
    for pose, verts in fr:                                         # freed blocks are handed out again
        ot.integrate(pose, verts)
        gt.integrate(pose, torch.from_numpy(verts).cuda())
    assert same_state(ot, gt) > n1
    gt.delete_blocks(torch.zeros((0, 4), dtype=torch.int32).cuda())
    assert gt.counters()["last_freed"] == 0


@pytest.mark.parametrize("variant", list(VARIANTS))
@pytest.mark.parametrize("threshold", [0.05, 2.0])
def test_garbage_collect_every_frame(oracle, vh, torch_cuda, variant, threshold):
    torch = torch_cuda
    ot, gt = pair(oracle, vh, variant, band=0.2, numBuckets=1 << 12, numVoxelBlocks=8192)
    freed = 0
    for pose, verts in frames(8):
        ot.integrate(pose, verts)
        gt.integrate(pose, torch.from_numpy(verts).cuda())
        want = ot.garbage_collect(threshold)
        gt.garbage_collect(threshold)
        same_state(ot, gt)
        c = gt.counters()
        assert c["last_freed"] == want and c["heap_exhausted"] == 0   # (who is refused on exhaustion is unspecified)
        freed += want
    assert gt.counters()["freed_total"] == freed
    if threshold == 0.05:
        assert freed > 500


def test_full_buckets_stay_prefixes(oracle, vh, torch_cuda):
    """8 buckets x 4 slots: deleting the first and third entry of a full bucket moves the others down."""
    torch = torch_cuda
    ot, gt = pair(oracle, vh, "fused-ballot-walk", numBuckets=8, bucketSize=4, numVoxelBlocks=256)
    for pose, verts in frames(12, step=2):
        ot.integrate(pose, verts)
        gt.integrate(pose, torch.from_numpy(verts).cuda())
    same_state(ot, gt, 4)
    tab = ot.hash_table().reshape(-1, 4).copy()
    victims = []
    for b in range(8):
        if (tab[b]["ptr"] != -1).all():
            victims += [tuple(tab[b]["pos"][0].tolist()), tuple(tab[b]["pos"][2].tolist())]
    assert victims
    ot.delete_blocks(victims)
    k4 = np.zeros((len(victims), 4), np.int32)
    k4[:, :3] = victims
    gt.delete_blocks(torch.from_numpy(k4).cuda())
    same_state(ot, gt, 4)
    # and the raycast still finds every remaining block (lookup relies on the prefix property)
    pose = frames(1)[0][0]
    out = torch.zeros((H, W), dtype=torch.float32, device="cuda")
    gt.raycast(pose, out)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), ot.raycast(pose))


def test_collection_on_shards(oracle, vh, torch_cuda):
    """Collection is bucket-local, so shards collect on their own: they stay equal to the slices of one
    unsharded oracle table that collects after the same multi-camera frames."""
    from test_sharding_cpu import check_shard_against_full
    torch = torch_cuda
    world = 2
    kw = dict(numBuckets=1 << 12, numVoxelBlocks=3000)
    plan = vdist.ShardPlan(kw["numBuckets"], world)
    shards = [vdist.HipShard(vh.default_params(**kw), W, H, 1, plan, r, W * H) for r in range(world)]
    full = oracle.OracleTable(oracle.default_params(**kw), W, H, 1)
    full.set_alloc_band(0.2)
    for sh in shards:
        sh.table.set_alloc_band(0.2)
    prims = synth.room_primitives()
    freed = 0
    for step in range(3):
        cams = []
        for r in range(world):
            pose = synth.camera_loop(60, phase=vdist.camera_phase(r, world))[(5 * step) % 60]
            cams.append((pose, synth.render_room_verts(pose, W, H, prims).numpy()))
        vdist.loopback_step(shards, [[c[0]] for c in cams], [[torch.from_numpy(c[1]).cuda()] for c in cams])
        vdist.reference_multi_camera_frame(full, [c[0] for c in cams], [c[1] for c in cams])
        # the unsharded table's compact list after the last camera's flatten holds only that camera's
        # blocks; collect what the shards saw for ALL cameras by listing those keys explicitly
        doomed = []
        for sh in shards:
            sh.table.synchronize()
            comp = sh.table.compact()
            vol = sh.table.sdf_blocks()
            for e in comp:
                v = vol[int(e["ptr"]):int(e["ptr"]) + 512]
                seen = v["weight"] > 0
                if not seen.any() or np.abs(v["sdf"][seen]).min() >= np.float32(0.05):
                    doomed.append(tuple(e["pos"].tolist()))
            sh.table.garbage_collect(0.05)
        freed += full.delete_blocks(doomed)
        assert len(doomed) == len(set(doomed))
    total = 0
    for r, sh in enumerate(shards):
        sh.table.synchronize()
        total += check_shard_against_full(sh.table, full, *plan.bucket_range(r), 5)
    assert total == len(full.allocated()) and freed > 200
    assert sum(sh.table.counters()["freed_total"] for sh in shards) == freed
