"""Corners of the C-ABI boundary on the GPU: the PtrContainer of the reference (SURVEY.md 8(a)
row T4, VoxelDataStructures.h:54-63) read back through raw device pointers, the candidate-list
overflow counter, option validation, and snapshot files that must not damage a live model."""
import ctypes as C
import struct

import numpy as np
import pytest

from conftest import entries_as_set
from voxelhashing_demo_amd import synth

pytestmark = pytest.mark.gpu
I4 = np.eye(4, dtype=np.float32)
KW = dict(numBuckets=1 << 17, numVoxelBlocks=4096)


def _hip_memcpy_d2h(vh, dev_ptr, nbytes):
    """Raw hipMemcpy (device -> host) through the HIP runtime the library itself is linked to:
    dlsym on the library's handle searches its dependency tree, so no second runtime is loaded."""
    from voxelhashing_demo_amd import _lib
    L = C.CDLL(_lib.LIB_PATH)
    L.hipMemcpy.restype = C.c_int
    L.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    buf = (C.c_uint8 * nbytes)()
    assert L.hipMemcpy(buf, C.c_void_p(dev_ptr), nbytes, 2) == 0          # hipMemcpyDeviceToHost
    return bytes(buf)


def test_ptr_container_layout_and_contents(vh, torch_cuda):
    """T4: 7 raw device pointers, 56 bytes, in the reference's order; each one addresses the
    buffer its name says (checked against vh_download / vh_get_counters)."""
    torch = torch_cuda
    from voxelhashing_demo_amd import _lib
    assert C.sizeof(_lib.PtrContainer) == 56
    assert [n for n, _ in _lib.PtrContainer._fields_] == [
        "d_heap", "d_hashTable", "d_compactifiedHashTable", "d_hashTableBucketMutex", "d_SDFBlocks",
        "d_heapCounter", "d_compactifiedHashCounter"]                      # VoxelDataStructures.h:55-62
    gt = vh.SDFHashtable(vh.default_params(**KW), 640, 480, 0)
    d_verts = torch.from_numpy(synth.sphere_inside_scene()).cuda()
    gt.integrate(I4, d_verts)
    gt.integrate(I4, d_verts)
    gt.synchronize()
    p = gt.device_pointers()
    ptrs = [getattr(p, n) for n, _ in _lib.PtrContainer._fields_]
    assert all(ptrs) and len(set(ptrs)) == 7
    c = gt.counters()
    assert struct.unpack("<i", _hip_memcpy_d2h(vh, p.d_heapCounter, 4))[0] == c["heap_counter"] == 4095 - 151
    assert struct.unpack("<i", _hip_memcpy_d2h(vh, p.d_compactifiedHashCounter, 4))[0] == c["occupied"] == 151
    table = gt.hash_table()
    n = 1 << 12
    raw = np.frombuffer(_hip_memcpy_d2h(vh, p.d_hashTable, 20 * n), dtype=vh.ENTRY_DTYPE)
    assert np.array_equal(raw, table[:n])
    first = int(np.nonzero(table["ptr"] != -1)[0][0])
    one = np.frombuffer(_hip_memcpy_d2h(vh, p.d_hashTable + 20 * first, 20), dtype=vh.ENTRY_DTYPE)[0]
    assert one == table[first] and one["ptr"] % 512 == 0
    comp = np.frombuffer(_hip_memcpy_d2h(vh, p.d_compactifiedHashTable, 20 * 151), dtype=vh.ENTRY_DTYPE)
    assert entries_as_set(comp) == entries_as_set(table[table["ptr"] != -1])
    heap = np.frombuffer(_hip_memcpy_d2h(vh, p.d_heap, 4 * 4096), dtype="<u4")
    assert np.array_equal(heap, gt.heap())
    vox = np.frombuffer(_hip_memcpy_d2h(vh, p.d_SDFBlocks + 8 * int(one["ptr"]), 4096), dtype=vh.VOXEL_DTYPE)
    assert np.array_equal(vox, gt.sdf_blocks()[int(one["ptr"]):int(one["ptr"]) + 512])
    # the bucket lock of the reference is an int per bucket; here an 8-byte word per bucket whose
    # upper half is the lock epoch: a bucket that received an entry in the last frame carries it
    h = first // 5
    word = struct.unpack("<Q", _hip_memcpy_d2h(vh, p.d_hashTableBucketMutex + 8 * h, 8))[0]
    assert 1 <= (word >> 55) <= c["epoch"] == 2            # top 9 bits: the lock epoch
    gt.close()


def test_candidate_overflow_is_counted_and_harmless(oracle, vh, torch_cuda):
    """A full candidate list drops contenders BEFORE they stake a claim: the loss is counted, the
    frame stays consistent, and the keys come back in the following frames."""
    torch = torch_cuda
    gt = vh.SDFHashtable(vh.default_params(**KW), 640, 480, 1)
    ot = oracle.OracleTable(oracle.default_params(**KW), 640, 480, 1)
    verts = synth.sphere_inside_scene()
    d_verts = torch.from_numpy(verts).cuda()
    gt.set_option("cand_capacity", 8)
    gt.integrate(I4, d_verts)
    c = gt.counters()
    assert c["cand_overflow"] > 0 and c["candidates"] > 8                 # demanded, not clamped
    assert 0 < c["allocated_total"] <= 8 and c["occupied"] == c["allocated_total"]
    prev = -1
    for _ in range(64):
        gt.integrate(I4, d_verts)
        n = gt.counters()["allocated_total"]
        if n == prev:
            break
        prev = n
    for _ in range(3):
        ot.integrate(I4, verts)
    tab = gt.hash_table()
    alloc = tab[tab["ptr"] != -1]
    assert entries_as_set(alloc) == entries_as_set(ot.allocated()) and len(alloc) == 179
    assert len(set(alloc["ptr"].tolist())) == len(alloc)
    gt.close()


def test_options_are_validated(vh, torch_cuda):
    gt = vh.SDFHashtable(vh.default_params(**KW), 640, 480, 1)
    for bad in (0, 1, 2, 5, 6, 7, -1):
        with pytest.raises(vh.VoxelHashError):
            gt.set_option("flatten_variant", bad)
    for good in (3, 4):
        gt.set_option("flatten_variant", good)
    with pytest.raises(vh.VoxelHashError):
        gt.set_option("no_such_option", 1)
    gt.close()


def test_bad_snapshots_leave_the_model_untouched(oracle, vh, torch_cuda, tmp_path):
    """vh_load_snapshot validates the whole file on the host before it changes device state."""
    torch = torch_cuda
    kw = dict(numBuckets=1 << 12, numVoxelBlocks=2048)
    poses = synth.camera_loop(500)
    prims = synth.room_primitives()
    frames = [(poses[i], synth.render_room_verts(poses[i], 320, 240, prims).numpy()) for i in (0, 3, 6)]
    ot = oracle.OracleTable(oracle.default_params(**kw), 320, 240, 1)
    gt = vh.SDFHashtable(vh.default_params(**kw), 320, 240, 1)
    for pose, v in frames[:2]:
        gt.integrate(pose, torch.from_numpy(v).cuda())
        ot.integrate(pose, v)
    good = tmp_path / "good.vhsnap"
    gt.save_snapshot(good)
    assert not (tmp_path / "good.vhsnap.partial").exists()                # written aside, renamed at the end
    blob = good.read_bytes()
    before_tab, before_vol, before_c = gt.hash_table(), gt.sdf_blocks(), gt.counters()
    n_entries = (1 << 12) * 5
    table_at = len(blob) - 4096 * len(gt.allocated()) - 4 * 2048 - 20 * n_entries
    heap_counter_at = 8 + 176 + 3 * 4 + 2 * 4                             # SnapshotHeader: magic, params, size, bucket range
    assert struct.unpack_from("<i", blob, heap_counter_at)[0] == before_c["heap_counter"]
    first_live = int(np.nonzero(before_tab["ptr"] != -1)[0][0])
    cases = {
        "truncated payload": blob[:-100],
        "truncated table": blob[:table_at + 1000],
        "trailing bytes": blob + b"\0" * 8,
        "bad magic": b"XXSNAP01" + blob[8:],
        "heap counter out of range": blob[:heap_counter_at] + struct.pack("<i", 1 << 20) + blob[heap_counter_at + 4:],
        "misaligned ptr": blob[:table_at + 20 * first_live + 12] + struct.pack("<i", 513) + blob[table_at + 20 * first_live + 16:],
        "ptr out of pool": blob[:table_at + 20 * first_live + 12] + struct.pack("<i", 512 * 4096) + blob[table_at + 20 * first_live + 16:],
    }
    for name, data in cases.items():
        bad = tmp_path / "bad.vhsnap"
        bad.write_bytes(data)
        with pytest.raises(vh.VoxelHashError):
            gt.load_snapshot(bad)
        assert np.array_equal(gt.hash_table(), before_tab), name
        assert np.array_equal(gt.sdf_blocks().view(np.uint32), before_vol.view(np.uint32)), name
        assert gt.counters()["heap_counter"] == before_c["heap_counter"], name
    # a context with another voxel size refuses the file (the kernels would disagree with the header)
    other = vh.SDFHashtable(vh.default_params(voxelSize=0.01, **kw), 320, 240, 1)
    with pytest.raises(vh.VoxelHashError, match="does not match"):
        other.load_snapshot(good)
    other.close()
    # ... and the live model still fuses on exactly like the oracle
    pose, v = frames[2]
    gt.integrate(pose, torch.from_numpy(v).cuda())
    ot.integrate(pose, v)
    gt.synchronize()
    from test_gpu_parity import _compare
    _compare(ot, gt)
    gt.close()


def test_batch_flush_and_fixed_slot_argument_checks(oracle, vh, torch_cuda):
    """vh_integrate_batch / vh_integrate_depth_batch / vh_flush / vh_export_views_fixed / vh_import_views:
    null and out-of-range arguments come back as VH_ERR_INVALID_ARGUMENT (never a crash), an empty batch
    and a flush with nothing pending are no-ops, and a failed call leaves the model as it was."""
    import ctypes as C
    torch = torch_cuda
    L = vh.load()
    gt = vh.SDFHashtable(vh.default_params(**KW), 640, 480, 1)
    ot = oracle.OracleTable(oracle.default_params(**KW), 640, 480, 1)
    verts = synth.sphere_inside_scene()
    d = torch.from_numpy(verts).cuda()
    h = gt._h
    pose = (C.c_float * 16)(*I4.reshape(-1))
    ptrs = (C.c_void_p * 1)(d.data_ptr())
    bad = 1                                                            # VH_ERR_INVALID_ARGUMENT (voxelhash.h)
    for rc in (L.vh_integrate_batch(None, 1, pose, ptrs, None), L.vh_integrate_batch(h, -1, pose, ptrs, None),
               L.vh_integrate_batch(h, 1, None, ptrs, None), L.vh_integrate_batch(h, 1, pose, None, None),
               L.vh_integrate_depth_batch(h, 1, pose, None, None), L.vh_flush(None),
               L.vh_export_views_fixed(h, None, 1, 0.1, 5.0, None, 16, None),
               L.vh_export_views_fixed(h, d.data_ptr(), 0, 0.1, 5.0, d.data_ptr(), 16, d.data_ptr()),
               L.vh_export_views_fixed(h, d.data_ptr(), 17, 0.1, 5.0, d.data_ptr(), 16, d.data_ptr()),
               L.vh_export_views_fixed(h, d.data_ptr(), 1, 5.0, 0.1, d.data_ptr(), 16, d.data_ptr()),
               L.vh_import_views(h, None, 1, 16, None), L.vh_import_views(h, d.data_ptr(), 0, 16, None),
               L.vh_import_views(h, d.data_ptr(), 1, 0, None)):
        assert rc == bad, rc
    assert L.vh_integrate_batch(h, 0, None, None, None) == 0          # an empty batch
    assert L.vh_flush(h) == 0                                          # nothing pending
    assert gt.counters()["allocated_total"] == 0 and gt.counters()["epoch"] == 0
    gt.integrate_batch([I4, I4], [d, d])
    ot.integrate(I4, verts)
    ot.integrate(I4, verts)
    assert entries_as_set(gt.allocated()) == entries_as_set(ot.allocated()) and len(gt.allocated()) == 179
    gt.close()


@pytest.mark.parametrize("pipelined", [0, 1])
def test_compact_table_is_dense_for_every_observer(oracle, vh, torch_cuda, pipelined):
    """Inside a fused frame the compact list has two ends (two counters instead of one hot word); whoever looks
    from outside gets the reference's dense list: the raw d_compactifiedHashTable pointer after vh_flush, a second
    look (nothing is folded twice), vh_download, and the step-level integrateDepthMap over the last frame's list
    (which the oracle replays as a second TSDF update of the same frame)."""
    import ctypes as C
    torch = torch_cuda
    from voxelhashing_demo_amd import _lib
    L = vh.load()
    hip = C.CDLL("libamdhip64.so")
    kw = dict(numBuckets=1 << 14, numVoxelBlocks=8192)
    W, H = 320, 240
    ot = oracle.OracleTable(oracle.default_params(**kw), W, H, 1)
    gt = vh.SDFHashtable(vh.default_params(**kw), W, H, 1)
    gt.set_option("pipeline", pipelined)
    prims, poses = synth.room_primitives(), synth.camera_loop(60)
    for i in (0, 3, 6, 9):
        v = synth.render_room_verts(poses[i], W, H, prims).numpy()
        gt.integrate(poses[i], torch.from_numpy(v).cuda())
        ot.integrate(poses[i], v)
    want = entries_as_set(ot.compact())
    assert len(want) > 200
    pc = _lib.PtrContainer()
    assert L.vh_get_device_pointers(gt._h, C.byref(pc)) == 0      # (flushes the pending frame and folds)
    torch.cuda.synchronize()
    n = gt.counters()["occupied"]
    assert n == len(want)
    for _ in range(2):
        raw = np.zeros(n, vh.ENTRY_DTYPE)
        assert L.vh_flush(gt._h) == 0
        torch.cuda.synchronize()
        assert hip.hipMemcpy(raw.ctypes.data_as(C.c_void_p), C.c_void_p(pc.d_compactifiedHashTable), raw.nbytes, 2) == 0
        assert entries_as_set(raw) == want and (raw["ptr"] != -1).all()
    assert entries_as_set(gt.compact()) == want
    # integrateDepthMap over that list: the last frame's TSDF update once more
    gt.integrate_depth_map(torch.from_numpy(v).cuda())
    ot.integrate_depth_map(v)
    gt.synchronize()
    gtab, otab = gt.hash_table(), ot.hash_table()
    assert np.array_equal(gtab["pos"], otab["pos"])
    gvol, ovol = gt.sdf_blocks(), ot.sdf_blocks()
    omap = {tuple(e["pos"]): int(e["ptr"]) for e in otab[otab["ptr"] != -1]}
    for e in gtab[gtab["ptr"] != -1][::5]:
        a = gvol[int(e["ptr"]):int(e["ptr"]) + 512]
        b = ovol[omap[tuple(e["pos"])]:omap[tuple(e["pos"])] + 512]
        assert np.array_equal(a["sdf"].view(np.uint32), b["sdf"].view(np.uint32)) and np.array_equal(a["weight"], b["weight"])
    gt.close()


def test_raw_pointers_fetched_once_stay_good_across_pipelined_frames(oracle, vh, torch_cuda):
    """The reference fetches its PtrContainer ONCE (deviceAllocate, VoxelUtils.cu:141-148).  Pipelined frames
    alternate between two compact / claim buffers internally; after ANY number of them (odd counts leave the
    frame's list in the second buffer) a vh_flush must leave the dense list behind the pointer fetched at the
    start, and vh_get_device_pointers must keep returning the same addresses."""
    import ctypes as C
    torch = torch_cuda
    from voxelhashing_demo_amd import _lib
    L = vh.load()
    hip = C.CDLL("libamdhip64.so")
    kw = dict(numBuckets=1 << 14, numVoxelBlocks=8192)
    W, H = 320, 240
    ot = oracle.OracleTable(oracle.default_params(**kw), W, H, 1)
    gt = vh.SDFHashtable(vh.default_params(**kw), W, H, 1)
    pc0 = _lib.PtrContainer()
    assert L.vh_get_device_pointers(gt._h, C.byref(pc0)) == 0          # before the first frame, as the reference does
    prims, poses = synth.room_primitives(), synth.camera_loop(60)
    frame = 0
    for count in (1, 3, 2, 1, 5):                                      # odd and even runs of pipelined frames
        ps, vs = [], []
        for _ in range(count):
            v = synth.render_room_verts(poses[3 * frame], W, H, prims).numpy()
            ot.integrate(poses[3 * frame], v)
            ps.append(poses[3 * frame])
            vs.append(torch.from_numpy(v).cuda())
            frame += 1
        gt.integrate_batch(ps, vs)                                     # (ends with vh_flush)
        torch.cuda.synchronize()
        want = entries_as_set(ot.compact())
        cnt = np.zeros(1, np.int32)
        assert hip.hipMemcpy(cnt.ctypes.data_as(C.c_void_p), C.c_void_p(pc0.d_compactifiedHashCounter), 4, 2) == 0
        assert int(cnt[0]) == len(want)
        raw = np.zeros(len(want), vh.ENTRY_DTYPE)
        assert hip.hipMemcpy(raw.ctypes.data_as(C.c_void_p), C.c_void_p(pc0.d_compactifiedHashTable), raw.nbytes, 2) == 0
        assert entries_as_set(raw) == want and (raw["ptr"] != -1).all(), f"after a run of {count} pipelined frames"
        pc = _lib.PtrContainer()
        assert L.vh_get_device_pointers(gt._h, C.byref(pc)) == 0
        for name, _ in _lib.PtrContainer._fields_:
            assert getattr(pc, name) == getattr(pc0, name), name
    # a two-launch frame and a step-level flatten after an odd pipelined run also end in the home buffer
    v = synth.render_room_verts(poses[3 * frame], W, H, prims).numpy()
    gt.integrate_batch([poses[3 * frame]], [torch.from_numpy(v).cuda()])
    ot.integrate(poses[3 * frame], v)
    gt.integrate(poses[3 * frame + 3], torch.from_numpy(v).cuda())
    ot.integrate(poses[3 * frame + 3], v)
    for step_level in (0, 1):
        if step_level:
            gt.set_pose(poses[3 * frame + 6]); ot.set_pose(poses[3 * frame + 6])
            gt.flatten(); ot.flatten()
        assert L.vh_flush(gt._h) == 0
        torch.cuda.synchronize()
        want = entries_as_set(ot.compact())
        raw = np.zeros(len(want), vh.ENTRY_DTYPE)
        assert hip.hipMemcpy(raw.ctypes.data_as(C.c_void_p), C.c_void_p(pc0.d_compactifiedHashTable), raw.nbytes, 2) == 0
        assert entries_as_set(raw) == want, f"step_level={step_level}"
    gt.close()


def test_compact_fold_on_a_nearly_full_table(oracle, vh, torch_cuda):
    """End B of the two-ended compact list is folded behind end A in place; on a small, dense table (more than
    two thirds of the entries visible) the two ends meet and source and destination of a naive fold overlap.
    2 560 entries = three walk tiles (ends A, B, A), 85 % of them allocated and visible: the dense list must hold
    each exactly once, in both frame forms."""
    torch = torch_cuda
    kw = dict(numBuckets=512, bucketSize=5, numVoxelBlocks=4096, voxelSize=0.005)
    ot = oracle.OracleTable(oracle.default_params(**kw), 640, 480, 1)
    gt = vh.SDFHashtable(vh.default_params(**kw), 640, 480, 1)
    verts = synth.sphere_inside_scene()
    d = torch.from_numpy(verts).cuda()
    for pipelined in (0, 1, 0, 1, 1, 0):
        for _ in range(2):                       # one insertion per bucket and frame: a few frames fill the table
            ot.integrate(I4, verts)
        if pipelined:
            gt.integrate_batch([I4] * 2, [d] * 2)
        else:
            for _ in range(2):
                gt.integrate(I4, d)
        got, want = gt.compact(), ot.compact()
        assert len(got) == len(want) and entries_as_set(got) == entries_as_set(want)
        assert len(entries_as_set(got)) == len(got), "an entry is listed twice"
    assert len(want) > 2 * 2560 // 3 + 300, len(want)
    gt.close()
