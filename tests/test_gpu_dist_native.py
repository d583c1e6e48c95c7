"""The multi-GPU host inside the library (include/voxelhash_dist.h: vh_dist_* on RCCL directly), with one rank on the GPU
box: the pipelined exchange (generate -> ncclAllToAll + ncclAllGather -> apply, three streams) must leave the oracle's
table, in both packet formats; vh_dist_raycast must render the oracle's image; and a plain C++ program drives it without
Python or torch (tests/cpp/sharded_demo.cpp).  N > 1: tests/test_gpu_dist_loopback.py (the same host code, ranks of one
process joined by the loop-back transport) and tests/test_gpu_dist_rccl.py (RCCL itself, one process per rank on the one GPU)."""
import os
import re
import subprocess

import numpy as np
import pytest

from test_sharding_cpu import check_shard_against_full
from voxelhashing_demo_amd import dist as vdist
from voxelhashing_demo_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H = 320, 240
KW = dict(numBuckets=1 << 14, numVoxelBlocks=1 << 13)


def _frames(n):
    poses = synth.camera_loop(120)[:n]
    prims = synth.room_primitives()
    return poses, [synth.render_room_verts(p, W, H, prims).numpy() for p in poses]


@pytest.mark.parametrize("sensor,forced", [(True, False), (False, False), (True, True), (False, True)])
def test_native_exchange_with_one_rank_equals_the_oracle(oracle, vh, torch_cuda, sensor, forced):
    """forced: vh_dist_set_option "force_collectives" -- a one-rank group applies its frames straight from the send buffers (no
    collective); forced, it runs ncclAllToAll + ncclAllGather all the same, through the receive buffers: the
    call sequence of an R-GPU node driven through vh_dist_step_batch without the NCCL_HOSTID rig.  Both: the start-up self-check
    of the transport, and the per-exchange phase times."""
    torch = torch_cuda
    batch, steps = 3, 5
    poses, verts = _frames(batch * steps)
    kinv = np.linalg.inv(synth.K_matrix(W, H).astype(np.float64)).astype(np.float32)
    ot = oracle.OracleTable(oracle.default_params(**KW), W, H, 1)
    if sensor:
        d16 = [np.round(v[..., 2] * 5000.0).clip(0, 65535).astype(np.uint16) for v in verts]
        verts = [oracle.preprocess(d, kinv)[0] for d in d16]
        frames = [torch.from_numpy(d).cuda() for d in d16]
    else:
        frames = [torch.from_numpy(v).cuda() for v in verts]
    torch.cuda.synchronize()
    nd = vdist.NativeDist(vh.default_params(**KW), W, H, 1, 0, 1, batch, vdist.unique_id(), sensor_k_inv=kinv if sensor else None)
    nd.self_check()                                     # a pattern through ncclAllToAll / ncclAllGather, compared on the device
    nd.set_option("force_collectives", 1 if forced else 0)
    nd.set_option("phase_timing", 1)
    for s in range(steps):
        k = s * batch
        nd.step(poses[k:k + batch], frames[k:k + batch])
        for j in range(k, k + batch):
            ot.integrate(poses[j], verts[j])            # one camera: the multi-camera frame is integrate()
    nd.flush()
    ph = nd.phase_times()
    assert ph["exchanges"] == steps and ph["generate"] > 0 and ph["apply"] > 0 and ph["first_to_last"] >= ph["apply"]
    # nothing is sent by a lone rank unless forced: the two timing events back to back (~5.5 us) against two RCCL kernels (15 us and more)
    assert (ph["collectives"] > 10.0) == forced, ph
    nd.set_option("phase_timing", 0)
    assert check_shard_against_full(nd.table, ot, 0, KW["numBuckets"], 5) > 100
    c = nd.table.counters()
    assert c["bin_overflow"] == 0 and c["epoch"] == batch * steps
    sec, calls = nd.host_stats()
    assert calls == steps and sec > 0
    # the raycast round: pose all-gather, export, all-to-all of record slots, import, raycast -- on the device throughout
    depth = torch.empty((H, W), dtype=torch.float32, device="cuda")
    lost = torch.zeros(1, dtype=torch.int32, device="cuda")
    for pose in (poses[2], poses[9], poses[2]):           # (back-to-back rounds: the pose travels by value)
        nd.raycast(pose, depth, 2048, lost=lost)
    nd.flush()
    torch.cuda.synchronize()
    want = ot.raycast(poses[2])
    assert np.array_equal(depth.cpu().numpy().view(np.uint32), want.view(np.uint32)) and (want > 0).mean() > 0.3
    assert int(lost.item()) == 0
    nd.raycast(poses[9], depth, 16, lost=lost)            # far too few record slots: reported, not fatal
    torch.cuda.synchronize()
    assert int(lost.item()) > 0
    nd.close()
    ot.close()


@pytest.mark.parametrize("batch", [1, 4, 8])
def test_fused_and_separate_generation_agree(oracle, vh, torch_cuda, batch):
    """Option "fused_generation" (the default): the key generation of exchange n rides in the frame launches that apply exchange
    n-2 (GenJob role, per-frame counters in the bins, no event operation with one rank).  The same frames through the fused form,
    through the separate launches, and through a run that switches between the two behind a flush must leave one table: the
    oracle's.  No timing events here, so the one-rank fused calls run in their quiet form."""
    torch = torch_cuda
    steps = 9
    poses, verts = _frames(min(120, batch * steps))
    steps = len(poses) // batch
    kinv = np.linalg.inv(synth.K_matrix(W, H).astype(np.float64)).astype(np.float32)
    d16 = [np.round(v[..., 2] * 5000.0).clip(0, 65535).astype(np.uint16) for v in verts]
    frames = [torch.from_numpy(d).cuda() for d in d16]
    torch.cuda.synchronize()
    ot = oracle.OracleTable(oracle.default_params(**KW), W, H, 1)
    for j in range(steps * batch):
        ot.integrate(poses[j], oracle.preprocess(d16[j], kinv)[0])
    for plan in ("fused", "separate", "switching", "walks"):
        nd = vdist.NativeDist(vh.default_params(**KW), W, H, 1, 0, 1, batch, vdist.unique_id(), sensor_k_inv=kinv)
        # (2 = wherever the launch can carry the role; the default, 1, adds a size rule that keeps tables as small as this test's on
        # the separate path: multi_fusing_pays)
        nd.table.set_option("flatten_variant", 3)     # (the reference's walk: the library's default, the walk-free launch, never carries the role)
        nd.set_option("fused_generation", 0 if plan == "separate" else 2)
        for s in range(steps):
            if plan == "switching" and s in (4, 6):     # (behind a flush; the first two calls after it generate separately)
                nd.flush()
                nd.set_option("fused_generation", 0 if s == 4 else 2)
            if plan == "walks" and s in (3, 6):         # (the walk-free launch cannot carry the role: the library changes its host
                nd.flush()                              #  path behind a flush, and back)
                nd.table.set_option("flatten_variant", 4 if s == 3 else 3)
            k = s * batch
            nd.step(poses[k:k + batch], frames[k:k + batch])
        nd.flush()
        assert check_shard_against_full(nd.table, ot, 0, KW["numBuckets"], 5) > 100, plan
        c = nd.table.counters()
        assert c["bin_overflow"] == 0 and c["epoch"] == batch * steps and c["spin_timeouts"] == 0, (plan, c)
        nd.close()
    ot.close()


@pytest.mark.parametrize("fused", [2, 0])
def test_key_bins_too_small_are_counted_not_overrun(oracle, vh, torch_cuda, fused):
    """A bin that cannot hold a batch's keys: the generator (the separate launches and the role of the frame launches alike, with its
    per-frame counters in the bin's last two records) drops what does not fit, the owner's launch counts the overflow and applies
    what arrived; nothing is written past a bin -- every allocated block is one the oracle has -- and the exchange goes on."""
    from conftest import entries_as_set
    torch = torch_cuda
    batch, steps = 4, 6
    poses, verts = _frames(batch * steps)
    kinv = np.linalg.inv(synth.K_matrix(W, H).astype(np.float64)).astype(np.float32)
    d16 = [np.round(v[..., 2] * 5000.0).clip(0, 65535).astype(np.uint16) for v in verts]
    frames = [torch.from_numpy(d).cuda() for d in d16]
    torch.cuda.synchronize()
    ot = oracle.OracleTable(oracle.default_params(**KW), W, H, 1)
    for j in range(batch * steps):
        ot.integrate(poses[j], oracle.preprocess(d16[j], kinv)[0])
    want = entries_as_set(ot.allocated())
    nd = vdist.NativeDist(vh.default_params(**KW), W, H, 1, 0, 1, batch, vdist.unique_id(), sensor_k_inv=kinv, key_capacity=96)
    nd.table.set_option("flatten_variant", 3)         # (so that fused = 2 does ride in the frame launches)
    nd.set_option("fused_generation", fused)
    for s in range(steps):
        k = s * batch
        nd.step(poses[k:k + batch], frames[k:k + batch])
    nd.flush()
    assert nd.generation_form() == ("fused" if fused else "separate")
    c = nd.table.counters()
    assert c["bin_overflow"] > 0 and c["spin_timeouts"] == 0 and c["epoch"] == batch * steps, c
    got = entries_as_set(nd.table.allocated())
    assert 0 < len(got) < len(want) and got <= want
    nd.close()
    ot.close()


def test_native_exchange_equals_the_python_pipeline(oracle, vh, torch_cuda):
    """Same frames through dist.ShardedPipeline (Python host, torch collectives) and through vh_dist_*: the same table."""
    import socket

    import torch.distributed as dist
    torch = torch_cuda
    batch, steps = 2, 4
    poses, verts = _frames(batch * steps)
    kinv = np.linalg.inv(synth.K_matrix(W, H).astype(np.float64)).astype(np.float32)
    d16 = [np.round(v[..., 2] * 5000.0).clip(0, 65535).astype(np.uint16) for v in verts]
    dd = [torch.from_numpy(d).cuda() for d in d16]
    dv = [torch.from_numpy(oracle.preprocess(d, kinv)[0]).cuda() for d in d16]
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        uid = vdist.unique_id(0, vdist.torch_broadcast_bytes())
        nd = vdist.NativeDist(vh.default_params(**KW), W, H, 1, 0, 1, batch, uid, sensor_k_inv=kinv, key_capacity=W * H // 4)
        plan = vdist.ShardPlan(KW["numBuckets"], 1)
        ts, fs = torch.cuda.Stream(), torch.cuda.Stream()
        sh = vdist.HipShard(vh.default_params(**KW), W, H, 1, plan, 0, W * H // 4, batch=batch, stream=ts, sets=2, sensor_k_inv=kinv)
        pipe = vdist.ShardedPipeline(sh, vdist.TorchDistTransport(), ts, fs)
        torch.cuda.synchronize()
        for s in range(steps):
            k = s * batch
            nd.step(poses[k:k + batch], dd[k:k + batch])
            pipe.feed(poses[k:k + batch], dv[k:k + batch], dd[k:k + batch])
        nd.flush()
        pipe.flush()
        a, b = nd.table.hash_table(), sh.table.hash_table()
        assert np.array_equal(a["pos"], b["pos"]) and np.array_equal(a["ptr"] != -1, b["ptr"] != -1) and (a["ptr"] != -1).sum() > 100
        for i in np.nonzero(a["ptr"] != -1)[0][::5]:
            assert np.array_equal(nd.table.block_voxels(a["ptr"][i]).view(np.uint32), sh.table.block_voxels(b["ptr"][i]).view(np.uint32))
        nd.close()
        sh.table.close()
    finally:
        dist.destroy_process_group()


def test_cpp_sharded_program(oracle, vh, torch_cuda, tmp_path):
    """tests/cpp/sharded_demo.cpp: a C++ host (no Python, no torch in that process) creates the communicator, feeds
    batches of sensor frames through SDF_Hashtable's multi-GPU constructor and prints what it built."""
    lib = os.path.join(ROOT, "voxelhashing_demo_amd", "lib")
    exe = tmp_path / "sharded_demo"
    subprocess.run(["/opt/rocm/bin/hipcc", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "sharded_demo.cpp"), "-o", str(exe),
                    "-L", lib, "-lsdf_hashtable", "-lvoxelhash_hip", f"-Wl,-rpath,{lib}"], check=True)
    batch, steps = 4, 3
    poses, verts = _frames(batch * steps)
    kinv = np.linalg.inv(synth.K_matrix(W, H).astype(np.float64)).astype(np.float32)
    d16 = [np.round(v[..., 2] * 5000.0).clip(0, 65535).astype(np.uint16) for v in verts]
    np.asarray(poses, np.float32).tofile(tmp_path / "poses.bin")
    np.stack(d16).tofile(tmp_path / "depth.bin")
    kinv.tofile(tmp_path / "kinv.bin")
    # (the program takes its HIP runtime and RCCL from the files this test session has already mapped -- torch's copies --
    # instead of paging the ROCm install's copies in from a cold disk: minutes on a fresh box)
    import torch
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0",
               LD_LIBRARY_PATH=tlib + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([str(exe), str(tmp_path / "poses.bin"), str(tmp_path / "depth.bin"), str(tmp_path / "kinv.bin"),
                          str(W), str(H), str(batch), str(steps), str(KW["numBuckets"]), str(KW["numVoxelBlocks"]),
                          str(tmp_path / "table.bin"), str(tmp_path / "depth_out.bin")],
                         check=True, capture_output=True, text=True, env=env).stdout
    got = dict((k, int(v)) for k, v in re.findall(r"(\w+)=(\d+)", out))
    ot = oracle.OracleTable(oracle.default_params(**KW), W, H, 1)
    for p, d in zip(poses, d16):
        ot.integrate(p, oracle.preprocess(d, kinv)[0])
    otab = ot.hash_table()
    assert got["allocated"] == int((otab["ptr"] != -1).sum()) > 100 and got["bin_overflow"] == 0
    tab = np.fromfile(tmp_path / "table.bin", dtype=vh.ENTRY_DTYPE)
    assert np.array_equal(tab["pos"], otab["pos"])
    depth = np.fromfile(tmp_path / "depth_out.bin", dtype=np.float32).reshape(H, W)
    assert np.array_equal(depth.view(np.uint32), ot.raycast(poses[5]).view(np.uint32))
    ot.close()


@pytest.mark.parametrize("world", [2, 4])
def test_cpp_threads_program_with_several_ranks(oracle, vh, torch_cuda, tmp_path, world):
    """tests/cpp/sharded_threads_demo.cpp: a C++ host (no Python, no torch in that process) with R ranks of SDF_Hashtable's
    multi-GPU constructor on the one GPU, one std::thread per rank, joined by the loop-back transport: every shard equals its
    slice of ONE oracle table, every rank's raycast through all shards equals the oracle's."""
    from voxelhashing_demo_amd import dist as vdist
    lib = os.path.join(ROOT, "voxelhashing_demo_amd", "lib")
    exe = tmp_path / "sharded_threads_demo"
    subprocess.run(["/opt/rocm/bin/hipcc", "-std=c++17", "-O1", "-pthread", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "sharded_threads_demo.cpp"), "-o", str(exe),
                    "-L", lib, "-lsdf_hashtable", "-lvoxelhash_hip", f"-Wl,-rpath,{lib}"], check=True)
    batch, steps = 2, 3
    n = batch * steps
    prims = synth.room_primitives()
    kinv = np.linalg.inv(synth.K_matrix(W, H).astype(np.float64)).astype(np.float32)
    cam_poses = [synth.camera_loop(60, phase=vdist.camera_phase(r, world))[:3 * n:3] for r in range(world)]
    d16 = [[np.round(synth.render_room_verts(p, W, H, prims).numpy()[..., 2] * 5000.0).clip(0, 65535).astype(np.uint16) for p in cam_poses[r]]
           for r in range(world)]
    np.asarray(cam_poses, np.float32).tofile(tmp_path / "poses.bin")
    np.stack([np.stack(d) for d in d16]).tofile(tmp_path / "depth.bin")
    kinv.tofile(tmp_path / "kinv.bin")
    import torch
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    env = dict(os.environ, LD_LIBRARY_PATH=tlib + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([str(exe), str(world), str(tmp_path / "poses.bin"), str(tmp_path / "depth.bin"), str(tmp_path / "kinv.bin"),
                          str(W), str(H), str(batch), str(steps), str(KW["numBuckets"]), str(KW["numVoxelBlocks"]), str(tmp_path) + "/"],
                         check=True, capture_output=True, text=True, env=env, timeout=600).stdout
    got = dict((k, int(v)) for k, v in re.findall(r"(\w+)=(\d+)", out))
    ot = oracle.OracleTable(oracle.default_params(**KW), W, H, 1)
    for j in range(n):
        vdist.reference_multi_camera_frame(ot, [cam_poses[r][j] for r in range(world)],
                                           [oracle.preprocess(d16[r][j], kinv)[0] for r in range(world)])
    otab = ot.hash_table()
    assert got["ranks"] == world and got["allocated"] == int((otab["ptr"] != -1).sum()) > 100 and got["bin_overflow"] == 0
    plan = vdist.ShardPlan(KW["numBuckets"], world)
    for r in range(world):
        lo, hi = plan.bucket_range(r)
        tab = np.fromfile(tmp_path / f"table{r}.bin", dtype=vh.ENTRY_DTYPE)
        assert np.array_equal(tab["pos"], otab["pos"][lo * 5:hi * 5]) and np.array_equal(tab["ptr"] != -1, otab["ptr"][lo * 5:hi * 5] != -1)
        depth = np.fromfile(tmp_path / f"depth{r}.bin", dtype=np.float32).reshape(H, W)
        assert np.array_equal(depth.view(np.uint32), ot.raycast(cam_poses[r][-1]).view(np.uint32)), r
    ot.close()
