"""The opt-in extensions of SURVEY.md 8(f) next #2 on the oracle (CPU): the overflow linked list the
reference carries as dead code (VoxelUtils.cu:384-411, 458-539, 578-602), the normal-directed block
DDA it has commented out (:632-703), and the two commented-out TSDF update variants (:815, :827).
These pin the SPEC (structure invariants, order independence, geometry); tests/test_gpu_overflow.py
holds the HIP path to the same oracle slot for slot."""
import numpy as np
import pytest

from oracle_shards import OracleShard
from voxelhashing_demo_amd import dist as vdist
from voxelhashing_demo_amd import synth

I4 = np.eye(4, dtype=np.float32)


def chains_ok(oracle, t):
    """Every allocated entry is reachable from its home bucket by the reference's lookup loop
    (:374-411): the bucket's slots, then at most attachedLinkedListSize iterations from its last slot."""
    tab = t.hash_table()
    p = t.params
    N, bs, L, nb = len(tab), p.bucketSize, p.attachedLinkedListSize, p.numBuckets
    lo = t.bucket_range[0]
    reached = set()
    for h in range(t.bucket_range[1] - lo):
        last = h * bs + bs - 1

        def home(e):
            return e["ptr"] != -1 and oracle.hash_block(*[int(c) for c in e["pos"]], nb) == h + lo
        for s in range(h * bs, h * bs + bs):
            if home(tab[s]):
                reached.add(s)
        i, it = last, 0
        while it < L:
            if home(tab[i]):
                reached.add(i)
            if tab[i]["offset"] == 0:
                break
            assert i == last or i % bs != bs - 1, "a chained entry sits in another bucket's head slot"
            i = (last + int(tab[i]["offset"])) % N
            it += 1
        else:
            pytest.fail(f"chain of bucket {h} is longer than the lookup loop reaches")
        if tab[last]["ptr"] == -1:
            assert tab[last]["offset"] == 0, "a free head slot must mean: no chain"
    alloc = set(np.nonzero(tab["ptr"] != -1)[0].tolist())
    assert reached == alloc
    keys = [tuple(k) for k in tab["pos"][sorted(alloc)].tolist()]
    assert len(set(keys)) == len(keys), "duplicate key"
    free = tab[tab["ptr"] == -1]
    assert np.all(free["offset"] == 0) and np.all(free["pos"] == 0x7fffffff)
    return len(alloc)


def converge(t, verts, frames=60):
    prev = -1
    for _ in range(frames):
        t.integrate(I4, verts)
        n = len(t.allocated())
        if n == prev:
            return n
        prev = n
    raise AssertionError("no convergence")


@pytest.mark.parametrize("nb,bs,L", [(48, 5, 6), (64, 4, 4), (96, 2, 8), (64, 2, 8)])
def test_overflow_list_places_what_the_bucket_drops(oracle, nb, bs, L):
    """The collision scene of G5 (151 block keys into a tiny table): without the list a full bucket
    drops the key for good; with it the key moves into a following bucket, one insertion per bucket
    and frame still holding for BOTH buckets involved."""
    verts = synth.sphere_inside_scene()
    kw = dict(numBuckets=nb, bucketSize=bs, numVoxelBlocks=1024, attachedLinkedListSize=L)
    plain = oracle.OracleTable(oracle.default_params(**kw), 640, 480, 0)
    lst = oracle.OracleTable(oracle.default_params(**kw), 640, 480, 0)
    lst.set_overflow(True)
    per_frame = []
    for f in range(60):
        before = len(lst.allocated())
        plain.integrate(I4, verts)
        lst.integrate(I4, verts)
        chains_ok(oracle, lst)
        after = len(lst.allocated())
        assert after - before <= nb                       # at most one insertion per bucket and frame
        per_frame.append(after - before)
        if after == before:
            break
    n_plain, n_list = len(plain.allocated()), len(lst.allocated())
    assert n_list > n_plain and n_list <= min(151, nb * bs)
    assert (lst.hash_table()["offset"] != 0).sum() >= n_list - n_plain > 0
    assert set(map(tuple, plain.allocated()["pos"].tolist())) <= set(map(tuple, lst.allocated()["pos"].tolist()))
    if (nb, bs, L) == (48, 5, 6):
        assert n_list == 150 and n_plain == 139           # one key stays out: its chain is at the loop's reach
    # the TSDF of a block does not depend on where its entry lives
    pv, lv = plain.sdf_blocks(), lst.sdf_blocks()
    lmap = {tuple(e["pos"]): int(e["ptr"]) for e in lst.allocated()}
    frames_seen = {}
    for e in plain.allocated()[::7]:
        a, b = pv[int(e["ptr"]):int(e["ptr"]) + 512], lv[lmap[tuple(e["pos"])]:lmap[tuple(e["pos"])] + 512]
        # same voxels touched; weights may differ by the frame in which the block arrived
        assert np.array_equal(a["weight"] > 0, b["weight"] > 0)
        frames_seen[tuple(e["pos"])] = 1
    assert frames_seen


def test_overflow_deletion_is_order_independent(oracle):
    """Deleting a set of keys -- heads with followers, chained entries, plain slots -- gives the same
    table whatever the order, and what is left is still a well-formed set of chains."""
    verts = synth.sphere_inside_scene()
    kw = dict(numBuckets=48, bucketSize=5, numVoxelBlocks=1024, attachedLinkedListSize=6)

    def build():
        t = oracle.OracleTable(oracle.default_params(**kw), 640, 480, 0)
        t.set_overflow(True)
        converge(t, verts)
        return t
    ref = build()
    tab = ref.hash_table()
    bs = 5
    heads_with_chain = [i for i in range(bs - 1, len(tab), bs) if tab[i]["ptr"] != -1 and tab[i]["offset"] != 0]
    chained = [i for i in np.nonzero(tab["ptr"] != -1)[0]
               if oracle.hash_block(*[int(c) for c in tab[i]["pos"]], 48) != i // bs]
    plain = [i for i in np.nonzero(tab["ptr"] != -1)[0] if i % bs == 1][:6]
    assert len(heads_with_chain) >= 3 and len(chained) >= 5
    doomed = sorted(set(heads_with_chain[:4] + chained[::2] + plain))
    keys = [tuple(int(c) for c in tab[i]["pos"]) for i in doomed]
    results = []
    rng = np.random.RandomState(3)
    for trial in range(4):
        t = build()
        order = list(keys) if trial == 0 else [keys[j] for j in rng.permutation(len(keys))]
        if trial == 1:
            order = list(reversed(keys))
        freed = t.delete_blocks(order + [(99, 99, 99)])            # an absent key is skipped
        assert freed == len(keys)
        n = chains_ok(oracle, t)
        assert n == len(ref.allocated()) - len(keys)
        tt = t.hash_table()
        results.append((tt["pos"].copy(), (tt["ptr"] != -1).copy(), tt["offset"].copy(), t.heap_counter()))
        # the freed blocks are zero again and back on the heap
        assert t.heap_counter() == 1023 - n
        left = set(map(tuple, t.allocated()["pos"].tolist()))
        assert not (left & set(keys))
        # fusing on brings the keys back
        assert converge(t, verts) == len(ref.allocated())
        chains_ok(oracle, t)
    for r in results[1:]:
        assert all(np.array_equal(a, b) for a, b in zip(results[0][:3], r[:3])) and r[3] == results[0][3]


def test_overflow_with_the_list_off_is_the_reference(oracle):
    """set_overflow(False) is bit for bit the table without the feature (the live reference path)."""
    verts = synth.sphere_inside_scene()
    kw = dict(numBuckets=64, bucketSize=2, numVoxelBlocks=1024)
    a = oracle.OracleTable(oracle.default_params(**kw), 640, 480, 0)
    b = oracle.OracleTable(oracle.default_params(**kw), 640, 480, 0)
    b.set_overflow(True)
    b.set_overflow(False)
    for _ in range(3):
        a.integrate(I4, verts)
        b.integrate(I4, verts)
    assert np.array_equal(a.hash_table(), b.hash_table())


@pytest.mark.parametrize("world", [2, 4])
def test_overflow_chains_stay_inside_a_shard(oracle, world):
    """Bucket-range shards: a chain wraps inside its shard's bucket range, and R shards equal ONE
    table whose chains wrap inside segments of the same size."""
    W, H = 160, 120
    kw = dict(numBuckets=512, bucketSize=2, numVoxelBlocks=2048, attachedLinkedListSize=8)
    plan = vdist.ShardPlan(kw["numBuckets"], world)
    shards = [OracleShard(oracle, oracle.default_params(**kw), W, H, 1, plan, r, W * H + 1) for r in range(world)]
    full = oracle.OracleTable(oracle.default_params(**kw), W, H, 1)
    full.set_overflow(True, plan.per_shard)
    for sh in shards:
        sh.table.set_overflow(True)
    prims = synth.room_primitives()
    for step in range(6):
        cams = []
        for r in range(world):
            pose = synth.camera_loop(40, phase=vdist.camera_phase(r, world))[(3 * step) % 40]
            cams.append((pose, synth.render_room_verts(pose, W, H, prims).numpy()))
        vdist.loopback_step(shards, [[c[0]] for c in cams], [[c[1]] for c in cams])
        vdist.reference_multi_camera_frame(full, [c[0] for c in cams], [c[1] for c in cams])
    ftab = full.hash_table()
    total = 0
    for r, sh in enumerate(shards):
        lo, hi = plan.bucket_range(r)
        mine = sh.table.hash_table()
        assert np.array_equal(mine["pos"], ftab["pos"][lo * 2:hi * 2])
        assert np.array_equal(mine["offset"], ftab["offset"][lo * 2:hi * 2])
        total += chains_ok(oracle, sh.table)
    assert total == len(full.allocated()) and (ftab["offset"] != 0).sum() > 3


# ---------------------------------------------------------------------------------------------
# normal-directed block DDA (VoxelUtils.cu:632-703)
# ---------------------------------------------------------------------------------------------
def _plane_scene(W=160, H=120, z=1.3, tilt=0.35):
    """A tilted plane seen by the identity camera: vertex map + normal map (camera frame)."""
    fx, fy, cx, cy = synth.intrinsics(W, H)
    u, v = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    n = np.array([np.sin(tilt), 0.0, -np.cos(tilt)])
    dx, dy = (u - cx) / fx, (v - cy) / fy
    depth = (n[2] * z) / (dx * n[0] + dy * n[1] + n[2])              # plane through (0,0,z) with normal n
    verts = synth.verts_from_depth(depth.astype(np.float32), W, H)
    normals = np.zeros((H, W, 4), np.float32)
    normals[..., :3] = n.astype(np.float32)
    normals[0, :, :3] = 0                                              # preProcess leaves the border without normals
    return verts, normals, n


def test_normal_dda_band_geometry(oracle):
    W, H, band = 160, 120, 0.2
    verts, normals, n = _plane_scene(W, H)
    kw = dict(numBuckets=1 << 14, numVoxelBlocks=1 << 14)
    surf = oracle.OracleTable(oracle.default_params(**kw), W, H, 1)
    dda = oracle.OracleTable(oracle.default_params(**kw), W, H, 1)
    dda.set_alloc_band(band, oracle.BAND_NORMAL_DDA)
    for _ in range(12):
        surf.integrate(I4, verts)
        dda.integrate(I4, verts, normals)
    s_keys = set(map(tuple, surf.allocated()["pos"].tolist()))
    d_keys = set(map(tuple, dda.allocated()["pos"].tolist()))
    assert s_keys <= d_keys and len(d_keys) > 2 * len(s_keys)
    # every block a dense sampling of the segments p -+ b*n hits (and that passes the frustum test) is there ...
    vs = np.float32(0.02)
    want = set()
    for y in range(1, H, 7):
        for x in range(0, W, 5):
            p = verts[y, x, :3].astype(np.float64)
            for s in np.linspace(-band, band, 41):
                q = (p + s * n).astype(np.float32)
                want.add(tuple(int(c) for c in oracle.world2block(q, float(vs))))
    visible = {k for k in want if dda.block_in_frustum(k)}
    missing = visible - d_keys
    assert len(missing) <= len(visible) // 200, f"{len(missing)} of {len(visible)} sampled band blocks are missing"
    # ... and nothing farther from the plane than the band plus a block diagonal
    centres = (np.array(sorted(d_keys), np.float64) * 8 + 3.5) * 0.02
    dist = np.abs((centres - np.array([0, 0, 1.3])) @ n)
    assert dist.max() <= band + 0.16 * np.sqrt(3) / 2 + 0.02
    # pixels without a normal demand their surface block only: the first image row brings no band blocks
    only_row0 = oracle.OracleTable(oracle.default_params(**kw), W, H, 1)
    only_row0.set_alloc_band(band, oracle.BAND_NORMAL_DDA)
    v0 = np.zeros_like(verts)
    v0[0] = verts[0]
    n0 = np.zeros_like(normals)
    ref0 = oracle.OracleTable(oracle.default_params(**kw), W, H, 1)
    for _ in range(6):
        only_row0.integrate(I4, v0, n0)
        ref0.integrate(I4, v0)
    assert np.array_equal(only_row0.hash_table()["pos"], ref0.hash_table()["pos"])


def test_ray_band_mode_is_unchanged(oracle):
    """BAND_RAY (round 1's samples along the viewing ray) is still the default and ignores normals."""
    verts, normals, _ = _plane_scene()
    kw = dict(numBuckets=1 << 12, numVoxelBlocks=1 << 13)
    a = oracle.OracleTable(oracle.default_params(**kw), 160, 120, 1)
    b = oracle.OracleTable(oracle.default_params(**kw), 160, 120, 1)
    a.set_alloc_band(0.2)
    b.set_alloc_band(0.2, oracle.BAND_RAY)
    for _ in range(3):
        a.integrate(I4, verts)
        b.integrate(I4, verts, normals)
    assert np.array_equal(a.hash_table(), b.hash_table())


# ---------------------------------------------------------------------------------------------
# the commented-out TSDF update variants (:815, :827)
# ---------------------------------------------------------------------------------------------
def test_depth_truncation_and_sample_weight(oracle):
    verts = synth.sphere_inside_scene()
    kw = dict(numBuckets=1 << 14, numVoxelBlocks=4096, truncation=0.04, truncScale=0.02, integrationWeightSample=10)
    base = oracle.OracleTable(oracle.default_params(**kw), 640, 480, 1)
    both = oracle.OracleTable(oracle.default_params(**kw), 640, 480, 1)
    both.set_integrate_flags(oracle.INT_DEPTH_TRUNCATION | oracle.INT_WEIGHT_SAMPLE)
    for _ in range(3):                                   # (frame 0 and 1 allocate, every frame updates what is there)
        base.integrate(I4, verts)
        both.integrate(I4, verts)
    assert np.array_equal(base.hash_table()["pos"], both.hash_table()["pos"])     # allocation is untouched
    bv, fv = base.sdf_blocks(), both.sdf_blocks()
    # the sphere has radius 2: depth ~ 2 m, so truncation = 0.04 + 0.02 * depth ~ 0.08 and the sample
    # weight max(10 * 1.5 * (1 - (depth - 0.5) / 4.5), 1) ~ 10
    assert abs(float(np.abs(bv["sdf"]).max()) - 0.04) < 1e-6
    assert 0.07 < float(np.abs(fv["sdf"]).max()) < 0.09
    w = fv["weight"][fv["weight"] > 0]
    assert 9.0 < w.min() and w.max() < 3 * 11.5 and (bv["weight"] > 0).sum() < (fv["weight"] > 0).sum()
    single = oracle.OracleTable(oracle.default_params(**kw), 640, 480, 1)
    single.set_integrate_flags(oracle.INT_WEIGHT_SAMPLE)
    single.integrate(I4, verts)
    one = single.sdf_blocks()["weight"]
    one = one[one > 0]                                                            # one frame: the sample weight itself
    assert len(one) > 10000 and one.max() < 11.5
    depth = 0.5 + 4.5 * (1.0 - one.astype(np.float64) / 15.0)                     # invert the weight formula
    assert 1.4 < depth.min() and depth.max() < 2.7


def test_ray_dda_band_geometry(oracle):
    """BAND_RAY_DDA (round 4): every block the segment of the viewing ray between depths z - b and z + b crosses.  A superset
    of what dense sampling of those segments finds (up to the frustum test), nothing farther from the surface along the ray
    than the band plus a block, the surface blocks included; tighter than the five ray samples of BAND_RAY, which reach +-16 cm."""
    W, H, band = 160, 120, 0.1
    verts, _, n = _plane_scene(W, H)
    kw = dict(numBuckets=1 << 14, numVoxelBlocks=1 << 14)
    surf = oracle.OracleTable(oracle.default_params(**kw), W, H, 1)
    ray = oracle.OracleTable(oracle.default_params(**kw), W, H, 1)
    dda = oracle.OracleTable(oracle.default_params(**kw), W, H, 1)
    ray.set_alloc_band(band, oracle.BAND_RAY)
    dda.set_alloc_band(band, oracle.BAND_RAY_DDA)
    for _ in range(12):
        surf.integrate(I4, verts)
        ray.integrate(I4, verts)
        dda.integrate(I4, verts)
    s_keys = set(map(tuple, surf.allocated()["pos"].tolist()))
    r_keys = set(map(tuple, ray.allocated()["pos"].tolist()))
    d_keys = set(map(tuple, dda.allocated()["pos"].tolist()))
    assert s_keys <= d_keys and len(d_keys) > 1.5 * len(s_keys)
    # (BAND_RAY's outermost samples sit at +-2 half-block steps = +-16 cm for a 10 cm band: the DDA's set is the tighter one)
    assert len(d_keys) < len(r_keys) and len(d_keys - r_keys) <= len(d_keys) // 10
    vs = np.float32(0.02)
    want = set()
    for y in range(1, H, 7):
        for x in range(0, W, 5):
            p = verts[y, x, :3].astype(np.float64)
            for s in np.linspace(p[2] - band, p[2] + band, 41):
                q = (p * (s / p[2])).astype(np.float32)
                want.add(tuple(int(c) for c in oracle.world2block(q, float(vs))))
    visible = {k for k in want if dda.block_in_frustum(k)}
    missing = visible - d_keys
    assert len(missing) <= len(visible) // 200, f"{len(missing)} of {len(visible)} sampled band blocks are missing"
    centres = (np.array(sorted(d_keys), np.float64) * 8 + 3.5) * 0.02
    dist = np.abs((centres - np.array([0, 0, 1.3])) @ n)
    assert dist.max() <= band + 0.16 * np.sqrt(3) / 2 + 0.02       # (the ray leans at most ~35 degrees off the plane normal here: closer than along the ray)
    # a surface closer to the camera than the band: the band begins at the surface point
    near = oracle.OracleTable(oracle.default_params(**kw), W, H, 1)
    near.set_alloc_band(0.5, oracle.BAND_RAY_DDA)
    v2 = verts.copy()
    v2[..., :3] *= np.float32(0.3 / 1.3)
    near.integrate(I4, v2)
    assert len(near.allocated()) > 0
