"""Raycast over bucket-range shards, CPU side (oracle): the blocks a view can touch are gathered
from the shards into a view table; raycasting that table must equal raycasting the unsharded
table bit for bit (DESIGN.md section 6 "raycast")."""
import numpy as np
import pytest
import torch

from oracle_shards import OracleShard, OracleViewTable
from voxelhashing_demo_amd import dist as vdist
from voxelhashing_demo_amd import synth

W, H = 160, 120
KW = dict(numBuckets=1 << 12, numVoxelBlocks=4096)


def build_scene(oracle, world, sem=1, steps=3):
    prims = synth.room_primitives()
    plan = vdist.ShardPlan(KW["numBuckets"], world)
    full = oracle.OracleTable(oracle.default_params(**KW), W, H, sem)
    shards = [OracleShard(oracle, oracle.default_params(**KW), W, H, sem, plan, r, W * H + 1)
              for r in range(world)]
    poses = []
    for step in range(steps):
        cams = []
        for r in range(world):
            pose = synth.camera_loop(60, phase=vdist.camera_phase(r, world))[(5 * step) % 60]
            cams.append((pose, synth.render_room_verts(pose, W, H, prims).numpy()))
        vdist.loopback_step(shards, [[c[0]] for c in cams], [[torch.from_numpy(c[1])] for c in cams])
        vdist.reference_multi_camera_frame(full, [c[0] for c in cams], [c[1] for c in cams])
        poses = [c[0] for c in cams]
    return plan, full, shards, poses


@pytest.mark.parametrize("world", [1, 3, 4])
def test_view_table_raycasts_like_the_unsharded_table(oracle, world):
    plan, full, shards, poses = build_scene(oracle, world)
    views = [OracleViewTable(oracle, oracle.default_params(**KW), W, H, 1) for _ in range(world)]
    depths = vdist.loopback_raycast(shards, views, poses, capacity=2048)
    hits = 0
    for r in range(world):
        ref = full.raycast(poses[r])
        assert np.array_equal(depths[r], ref)
        hits += int((ref > 0).sum())
    assert hits > 1000 * world


def test_selection_is_a_superset_of_the_sampled_blocks(oracle):
    """Every allocated block a ray of the view samples is in the export (so the view table answers
    every lookup the raycast makes like the whole table would)."""
    plan, full, shards, poses = build_scene(oracle, 1)
    pose = poses[0]
    rec, n = shards[0].table.export_view(pose, 4096)
    assert n == len(rec)
    exported = {tuple(r[:12].view(np.int32)) for r in rec}
    allocated = {tuple(k) for k in full.allocated()["pos"].tolist()}
    assert exported <= allocated and 0 < len(exported) < len(allocated)
    # sample the rays the way the raycast does and collect the blocks they visit
    fx, fy, cx, cy = synth.intrinsics(W, H)
    T = np.asarray(pose, np.float32).reshape(4, 4)
    vs = np.float32(0.02)
    us, vv = np.meshgrid(np.arange(0, W, 3, dtype=np.float32), np.arange(0, H, 3, dtype=np.float32))
    dx, dy = (us - np.float32(cx)) / np.float32(fx), (vv - np.float32(cy)) / np.float32(fy)
    touched = set()
    for i in range(0, 246):
        t = np.float32(0.1) + np.float32(i) * vs
        pc = np.stack([dx * t, dy * t, np.full_like(dx, t), np.ones_like(dx)], -1)
        pw = pc @ T.T
        vox = np.trunc(pw[..., :3] / vs + np.copysign(np.float32(0.5), pw[..., :3])).astype(np.int64)
        touched |= {tuple(k) for k in np.floor_divide(vox, 8).reshape(-1, 3).tolist()}
    assert (touched & allocated) <= exported
    assert len(touched & allocated) > 50


def test_capacity_overflow_is_reported(oracle):
    plan, full, shards, poses = build_scene(oracle, 1, steps=1)
    rec, n = shards[0].table.export_view(poses[0], 5)
    assert n > 5 and len(rec) == 5


def test_reference_semantics_tables_raycast_too(oracle):
    plan, full, shards, poses = build_scene(oracle, 2, sem=0)
    views = [OracleViewTable(oracle, oracle.default_params(**KW), W, H, 0) for _ in range(2)]
    depths = vdist.loopback_raycast(shards, views, poses, capacity=2048)
    for r in range(2):
        assert np.array_equal(depths[r], full.raycast(poses[r]))
