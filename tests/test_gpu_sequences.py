"""Random sequences of API calls on one context against the oracle: frames in both launch forms and
several walk variants, pipelined frames and batches, step-level frames, collections, deletions, raycasts,
snapshot round trips.
State that survives between calls (lock epoch, counter parity, which counter holds the occupied
count, the armed compact counter) must never depend on what was called before."""
import numpy as np
import pytest

from conftest import blocks_by_pos
from voxelhashing_demo_amd import synth

pytestmark = pytest.mark.gpu

W, H = 160, 120


def same(ot, gt):
    gt.synchronize()
    a, b = ot.hash_table(), gt.hash_table()
    assert np.array_equal(a["pos"], b["pos"]) and np.array_equal(a["ptr"] != -1, b["ptr"] != -1)
    oa = blocks_by_pos(a[a["ptr"] != -1], ot.sdf_blocks())
    ga = blocks_by_pos(b[b["ptr"] != -1], gt.sdf_blocks())
    for k in oa:
        assert np.array_equal(oa[k].view(np.uint32), ga[k].view(np.uint32)), k
    c = gt.counters()
    assert c["heap_counter"] == ot.heap_counter() and c["heap_exhausted"] == 0
    assert c["occupied"] == ot.compact_count()
    return c


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("sem", [0, 1])
def test_random_call_sequences(oracle, vh, torch_cuda, tmp_path, seed, sem):
    torch = torch_cuda
    rng = np.random.RandomState(100 * sem + seed)
    kw = dict(numBuckets=257, bucketSize=4, numVoxelBlocks=4096)        # small, odd bucket count: crowded buckets
    ot = oracle.OracleTable(oracle.default_params(**kw), W, H, sem)
    gt = vh.SDFHashtable(vh.default_params(**kw), W, H, sem)
    prims = synth.room_primitives()
    poses = synth.camera_loop(40)
    frames = [(p, synth.render_room_verts(p, W, H, prims).numpy()) for p in poses[::4]]
    if sem == 0:
        frames = [(np.eye(4, dtype=np.float32), synth.sphere_inside_scene(W, H))] + frames[:3]
    depth = torch.zeros((H, W), device="cuda")
    log = []
    for step in range(45):
        op = rng.choice(["frame", "frame", "frame", "batch", "steps", "gc", "delete", "raycast", "snapshot", "option",
                         "pipeline", "band"])
        log.append(op)
        pose, verts = frames[rng.randint(len(frames))]
        if op == "frame":
            ot.integrate(pose, verts)
            gt.integrate(pose, torch.from_numpy(verts).cuda())
        elif op == "batch":                                             # several frames, one launch each + a flush
            part = [frames[rng.randint(len(frames))] for _ in range(int(rng.randint(1, 5)))]
            gt.integrate_batch([p for p, _ in part], [torch.from_numpy(v).cuda() for _, v in part])
            for p, v in part:
                ot.integrate(p, v)
        elif op == "pipeline":                                          # plain frames pipelined from here on (or not)
            gt.set_option("pipeline", int(rng.randint(2)))
        elif op == "steps":                                             # the reference's four calls, one by one
            d = torch.from_numpy(verts).cuda()
            for t, v in ((ot, verts), (gt, d)):
                t.set_pose(pose)
                t.reset_mutexes()
                t.alloc_blocks(v)
                t.flatten()
                t.integrate_depth_map(v)
        elif op == "gc":
            th = float(rng.choice([0.0, 0.03, 0.3, 5.0]))
            assert ot.garbage_collect(th) == (gt.garbage_collect(th), gt.counters()["last_freed"])[1]
        elif op == "delete":
            alloc = ot.allocated()["pos"]
            if len(alloc):
                keys = alloc[rng.choice(len(alloc), size=min(len(alloc), 7), replace=False)]
                ot.delete_blocks([tuple(k) for k in keys.tolist()])
                k4 = np.zeros((len(keys), 4), np.int32)
                k4[:, :3] = keys
                gt.delete_blocks(torch.from_numpy(k4).cuda())
        elif op == "raycast":
            gt.raycast(pose, depth)
            torch.cuda.synchronize()
            assert np.array_equal(depth.cpu().numpy().view(np.uint32), ot.raycast(pose).view(np.uint32)), log
        elif op == "snapshot":                                          # round trip through a file: nothing changes
            path = tmp_path / f"s{step}.bin"
            gt.save_snapshot(path)
            gt.load_snapshot(path)
            ot.delete_blocks([])                                        # a loaded model has no compact list yet
        elif op == "option":
            gt.set_option("fused_frame", int(rng.randint(2)))
            gt.set_option("flatten_variant", int(rng.choice([3, 4])))
            gt.set_option("walk_nt", int(rng.randint(2)))              # (on by default only beyond 256 MiB of table)
        elif op == "band":
            b = float(rng.choice([0.0, 0.1]))
            ot.set_alloc_band(b)
            gt.set_alloc_band(b)
        if op == "frame" and rng.randint(2):
            continue                                                    # (pipelined frames stay in flight across ops)
        if op not in ("raycast", "option", "band", "pipeline"):
            try:
                c = same(ot, gt)
            except AssertionError as e:
                raise AssertionError(f"after {log}: {e}")
    assert len(ot.allocated()) > 0
