#!/usr/bin/env python3
"""bench.py -- frames/s of the voxel-hashing TSDF path on MI355X (BASELINE.json metric).

A step = one 640x480 depth frame taken through SDF_Hashtable::integrate
(lock epoch -> allocBlocks -> flattenIntoBuffer -> integrateDepthMap) on the
synthetic room of config C2; vertex maps and poses are resident in HBM before
the timed region.  One JSON line on stdout (rank 0).

  python bench.py [--gpus N] [--steps K] [--warmup W]
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the pool's host driver only supports dmabuf IPC (RCCL / cross-process tensor sharing)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)

WORKLOADS = {
    # BASELINE.json configs[1]
    "C2": dict(width=640, height=480, frames=500, buckets=1 << 20, blocks=1 << 18, voxel=0.02,
               desc="C2: synthetic 6x3x5 m room, 640x480 x 500-pose camera loop, 2^20 buckets x 5, "
                    "2^18 voxel blocks, voxel 0.02 m, PINHOLE semantics"),
    # BASELINE.json configs[2] (HBM-bound stress); selectable with --workload C3
    "C3": dict(width=1280, height=960, frames=200, buckets=1 << 22, blocks=1 << 21, voxel=0.005,
               desc="C3: synthetic room, 1280x960, 2^22 buckets x 5, 2^21 voxel blocks, voxel 0.005 m, "
                    "PINHOLE semantics (200 distinct frames of the 2000-pose path resident)"),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default="C2", choices=sorted(WORKLOADS))
    ap.add_argument("--frames", type=int, default=0, help="distinct resident frames (default: workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=0, help="frames in the CPU sample (0 = auto, about 15 s)")
    ap.add_argument("--raycast-steps", type=int, default=50)
    ap.add_argument("--profile-steps", type=int, default=200)
    ap.add_argument("--batch", type=int, default=8,
                    help="sharded path: frames per camera carried by one all-to-all / all-gather")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="sharded path: do not overlap key generation + RCCL with the table work")
    ap.add_argument("--float-packets", action="store_true",
                    help="sharded path: float vertex maps and float camera-z packets instead of uint16 sensor depth")
    ap.add_argument("--sharded-raycast", action="store_true",
                    help="sharded path: also time the raycast over the shards (always on with one rank)")
    ap.add_argument("--sharded", action="store_true",
                    help="force the bucket-range-sharded path (torch.distributed) even with one rank")
    return ap.parse_args()


def baseline_metric(width, height):
    """BASELINE.json's metric string for the 640x480 workloads (value = the frames/s half; the raycast
    Mpix/s half travels in `raycast_mpix_per_s`), a plain description otherwise."""
    if (width, height) == (640, 480):
        try:
            return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
        except Exception:
            return "frames/s TSDF-integrated + raycast Mpix/s, 640\u00d7480, 1/2/4/8 MI355X"
    return f"frames/s TSDF-integrated + raycast Mpix/s, {width}x{height}"


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    import voxelhashing_demo_amd as V
    from voxelhashing_demo_amd import _lib as L
    from voxelhashing_demo_amd import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    if world > 1 or args.sharded:
        if "MASTER_ADDR" not in os.environ:
            import socket
            with socket.socket() as sk:              # direct (not torchrun) launch with --sharded: any free port
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
        # one node by contract: keep RCCL's bootstrap and the c10d store on the loop-back interface
        # (the container's hostname may not resolve), and surface a stuck collective in minutes
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        import datetime
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank),
                                timeout=datetime.timedelta(seconds=300))
        from voxelhashing_demo_amd import dist as vdist
        args.metric_name = baseline_metric(WORKLOADS[args.workload]["width"], WORKLOADS[args.workload]["height"])
        return vdist.bench_sharded(args, WORKLOADS[args.workload], rank, world, local_rank)

    wl = WORKLOADS[args.workload]
    Wd, Ht = wl["width"], wl["height"]
    nframes = args.frames or wl["frames"]
    dev = torch.device("cuda", local_rank)
    stream = torch.cuda.Stream(device=dev)

    # ---- inputs: rendered on the GPU, resident before timing ----
    poses = synth.camera_loop(wl["frames"])[:nframes]
    prims = synth.room_primitives()
    verts = torch.empty((nframes, Ht, Wd, 4), dtype=torch.float32, device=dev)
    for i in range(nframes):
        verts[i] = synth.render_room_verts(poses[i], Wd, Ht, prims, device=dev)
    torch.cuda.synchronize()

    params = V.default_params(numBuckets=wl["buckets"], numVoxelBlocks=wl["blocks"], voxelSize=wl["voxel"])
    table = V.SDFHashtable(params, Wd, Ht, V.SEM_PINHOLE, device=local_rank, stream=stream)
    lib, h = table._lib, table._h
    pose_ptrs = [np.ascontiguousarray(p.reshape(16)).ctypes.data_as(C.POINTER(C.c_float)) for p in poses]
    pose_keep = [np.ascontiguousarray(p.reshape(16)) for p in poses]
    pose_ptrs = [p.ctypes.data_as(C.POINTER(C.c_float)) for p in pose_keep]
    vert_ptrs = [verts[i].data_ptr() for i in range(nframes)]

    def step(i):
        k = i % nframes
        rc = lib.vh_integrate(h, pose_ptrs[k], vert_ptrs[k], None)
        if rc != 0:
            L.check(rc, "vh_integrate")

    # A full collection of the Python garbage collector walks every object torch has created (~10^6):
    # 30-50 ms, i.e. longer than the whole timed region.  Existing objects are moved out of its reach
    # and it stays off while the clock runs (the loop allocates only small short-lived objects).
    import gc
    gc.collect()
    gc.freeze()
    gc.disable()
    for i in range(500):           # fixed, untimed run-in (one lap of the loop) before the W warm-up steps
        step(i)
    table.synchronize()
    for i in range(args.warmup):
        step(i)
    table.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    table.synchronize()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    fps = args.steps / elapsed
    counters = table.counters()

    # ---- per-kernel durations: HIP events on the path's own stream ----
    nprof = args.profile_steps
    table.set_profiling(True)
    occ_sum = 0
    for i in range(nprof):
        step(args.warmup + args.steps + i)
    kt = table.kernel_times(reset=True)
    table.set_profiling(False)
    occ = table.counters()["occupied"]
    n_entries = table.num_entries
    fused = kt["frame_scan_claim_ms"] > 0
    if fused:
        # dominant kernel of the fused frame: per-pixel claim phase || walk over the VoxelEntry
        # array.  Algorithmic bytes of one launch (SURVEY.md 8(d) terms): the vertex map read once
        # by the claim half (16*W*H), one pass over the table (20*N), the compact entries written
        # (20*occ) and one 100-byte bucket probe per distinct block key (keys ~ occ).
        kname = "frame_scan_claim_kernel"
        dom_us = 1e3 * kt["frame_scan_claim_ms"] / max(1, kt["launches"])
        dom_bytes = 16 * Wd * Ht + 20 * n_entries + 20 * occ + 100 * occ
    else:
        # algorithmic bytes of one flatten launch: one pass over the VoxelEntry array + the
        # compact entries written
        kname = "flatten_kernel"
        dom_us = 1e3 * kt["flatten_ms"] / max(1, kt["launches"])
        dom_bytes = 20 * n_entries + 20 * occ
    achieved = dom_bytes / (dom_us * 1e-6) / 1e9
    traffic = None
    pmc_file = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if os.path.exists(pmc_file):
        try:
            pmc = json.load(open(pmc_file)).get(args.workload, {})
            # template argument 3 = the default ballot walk (WalkKind in vh_kernels.hip)
            traffic = pmc.get(kname + "<3>_hbm_bytes_per_launch", pmc.get(kname + "_hbm_bytes_per_launch"))
        except Exception:
            traffic = None
    roofline = dict(bound="hbm", kernel=kname, achieved=round(achieved, 1), peak=HBM_PEAK_GBS,
                    unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic,
                    bytes_per_launch=dom_bytes, us_per_launch=round(dom_us, 2))
    kernels_us = {k[:-3] + "_us": round(1e3 * v / max(1, kt["launches"]), 2)
                  for k, v in kt.items() if k.endswith("_ms") and k != "raycast_ms" and v > 0}
    # algorithmic bytes of the whole frame (SURVEY.md 8(d) B_frame; no mutex memset in this build)
    keys = occ   # distinct in-frustum block keys of a frame ~ occupied blocks (not counted on the device)
    b_frame = 16 * Wd * Ht + 4 * Wd * Ht + 20 * n_entries + 20 * occ + occ * (20 + 4096 + 4096) + 100 * keys
    frame_gbs = b_frame * fps / 1e9

    # ---- same workload with the opt-in occupancy-index walk (NOT the reference algorithm: the
    # flatten step reads the 1-bit-per-bucket index and only the non-empty buckets instead of
    # every VoxelEntry; reported separately, never as `value`) ----
    table.set_option("flatten_variant", 4)
    for i in range(args.warmup):
        step(i)
    table.synchronize()
    t2 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    table.synchronize()
    idx_elapsed = time.perf_counter() - t2
    table.set_option("flatten_variant", 3)
    index_variant = dict(value=round(args.steps / idx_elapsed, 1), unit="frames/s",
                         ms_per_step=round(1e3 * idx_elapsed / args.steps, 5),
                         flatten_bytes=wl["buckets"] // 8 + 100 * counters["allocated_total"],
                         note="vh_set_option(flatten_variant=4): walk over the bucket-occupancy bitmap "
                              "(numBuckets/8 bytes) + the non-empty buckets instead of the 20*N-byte table walk")

    # ---- the same frames straight from uint16 sensor depth (vh_integrate_depth: preProcess's vertex
    # computation inside the claim phase, no vertex map in memory), against the two-call form
    # vh_preprocess + vh_integrate.  Extension of the boundary; reported separately. ----
    k_inv = np.linalg.inv(synth.K_matrix(Wd, Ht).astype(np.float64)).astype(np.float32)
    nd = min(nframes, 250)
    depth16 = torch.empty((nd, Ht, Wd), dtype=torch.uint16, device=dev)
    for i in range(nd):
        depth16[i] = (verts[i, :, :, 2] * 5000.0).round().clamp(0, 65535).to(torch.uint16)
    kin = np.ascontiguousarray(k_inv.reshape(9))
    kin_p = kin.ctypes.data_as(C.POINTER(C.c_float))
    d_ptrs = [depth16[i].data_ptr() for i in range(nd)]
    tmp_v, tmp_n = torch.empty_like(verts[0]), torch.empty_like(verts[0])
    tv, tn, st = tmp_v.data_ptr(), tmp_n.data_ptr(), C.c_void_p(stream.cuda_stream)
    torch.cuda.synchronize()       # the images were written on torch's default stream, the table reads them on its own

    def fused_depth(i):
        lib.vh_integrate_depth(h, pose_ptrs[i % nd], d_ptrs[i % nd], kin_p)

    def two_calls(i):
        lib.vh_preprocess(d_ptrs[i % nd], kin_p, Wd, Ht, tv, tn, st)
        lib.vh_integrate(h, pose_ptrs[i % nd], tv, None)

    sensor = {}
    for name, fn in (("vh_integrate_depth", fused_depth), ("vh_preprocess + vh_integrate", two_calls)):
        for i in range(args.warmup):
            fn(i)
        table.synchronize()
        t3 = time.perf_counter()
        for i in range(args.steps):
            fn(args.warmup + i)
        table.synchronize()
        sensor[name] = round(args.steps / (time.perf_counter() - t3), 1)
    sensor_depth = dict(frames_per_s=sensor, unit="frames/s",
                        note="input: uint16 depth images (2 B/pixel); the fused call computes the vertices inside "
                             "the claim phase and gathers depth from the image, bit-equal to the two-call form")

    # ---- raycast Mpix/s (second half of the metric) ----
    depth = torch.empty((Ht, Wd), dtype=torch.float32, device=dev)
    dptr = depth.data_ptr()
    for i in range(5):
        lib.vh_raycast(h, pose_ptrs[i % nframes], 0.1, 5.0, dptr)
    table.synchronize()
    t1 = time.perf_counter()
    for i in range(args.raycast_steps):
        lib.vh_raycast(h, pose_ptrs[(7 * i) % nframes], 0.1, 5.0, dptr)
    table.synchronize()
    ray_s = time.perf_counter() - t1
    raycast_mpix = args.raycast_steps * Wd * Ht / ray_s / 1e6 if args.raycast_steps else None

    # ---- kernel time of the raycast and of the next-row entry points (all timed regions are over) ----
    table.set_profiling(True)
    for i in range(5):
        lib.vh_raycast(h, pose_ptrs[(7 * i) % nframes], 0.1, 5.0, dptr)
    kt_r = table.kernel_times(reset=True)
    raycast_us = 1e3 * kt_r["raycast_ms"] / max(1, kt_r["raycast_launches"])
    # SURVEY.md 8(d) raycast work unit: every visible block and the output touched once
    ray_bytes = (4096 + 20) * occ + 4 * Wd * Ht
    raycast = dict(mpix_per_s=round(raycast_mpix, 1) if raycast_mpix else None, kernel_us=round(raycast_us, 2),
                   kernel_mpix_per_s=round(Wd * Ht / raycast_us, 1) if raycast_us > 0 else None,
                   algorithmic_bytes=ray_bytes,
                   achieved_gbs=round(ray_bytes / (raycast_us * 1e-6) / 1e9, 1) if raycast_us > 0 else None,
                   bound="valu issue (5.6k VALU instructions per wave against 66 memory reads, DESIGN.md 4.1): "
                         "the byte rate is far below HBM by construction")
    # depth pre-processing (next #1): uint16 depth -> vertex + normal maps, one fused kernel
    depth_u16 = (verts[0, :, :, 2] * 5000.0).clamp(0, 65535).to(torch.uint16)
    pos_out, nrm_out = torch.empty_like(verts[0]), torch.empty_like(verts[0])
    torch.cuda.synchronize()
    k_inv = np.linalg.inv(synth.K_matrix(Wd, Ht).astype(np.float64)).astype(np.float32)
    with torch.cuda.stream(stream):
        for i in range(3):
            V.preprocess(depth_u16, k_inv, pos_out, nrm_out, stream=stream)
        stream.synchronize()
        t3 = time.perf_counter()
        for i in range(50):
            V.preprocess(depth_u16, k_inv, pos_out, nrm_out, stream=stream)
        stream.synchronize()
        pre_us = 1e6 * (time.perf_counter() - t3) / 50
    pre_bytes = (2 + 16 + 16) * Wd * Ht
    # camera tracking (next #4): one fused ICP round (pairing + Jacobian + 27 sums) and a raycast target
    from voxelhashing_demo_amd import tracking
    Kf = synth.K_matrix(Wd, Ht)
    trk = tracking.CameraTracking(Wd, Ht, Kf, stream=stream, flags=3)
    tgt_p, tgt_n = torch.empty_like(verts[0]), torch.empty_like(verts[0])
    with torch.cuda.stream(stream):
        lib.vh_raycast(h, pose_ptrs[0], 0.1, 5.0, dptr)
        tracking.depth_to_maps(depth, k_inv, tgt_p, tgt_n, stream=stream)
        trk.build_system(verts[1], tgt_p, tgt_n, np.eye(4))
        t4 = time.perf_counter()
        for i in range(20):
            icp_sys = trk.build_system(verts[1], tgt_p, tgt_n, np.eye(4))
        icp_us = 1e6 * (time.perf_counter() - t4) / 20
        trk.Align(verts[1], tgt_p, tgt_n)
        t5 = time.perf_counter()
        for i in range(5):
            trk.Align(verts[1], tgt_p, tgt_n)
        align_us = 1e6 * (time.perf_counter() - t5) / 5
    icp_bytes = 48 * Wd * Ht
    # garbage collection (next #4) over the blocks the last frame saw; threshold 0 frees them all
    step(0)
    table.garbage_collect(0.0)
    kt_g = table.kernel_times(reset=True)
    gc_counters = table.counters()
    table.set_profiling(False)
    # block silhouettes (row R1): front / back cube depth per pixel
    sil_f, sil_b = torch.empty((Ht, Wd), dtype=torch.float32, device=dev), torch.empty((Ht, Wd), dtype=torch.float32, device=dev)
    for i in range(3):
        lib.vh_render_blocks(h, pose_ptrs[0], 0.1, 5.0, sil_f.data_ptr(), sil_b.data_ptr())
    table.synchronize()
    t6 = time.perf_counter()
    for i in range(20):
        lib.vh_render_blocks(h, pose_ptrs[(7 * i) % nframes], 0.1, 5.0, sil_f.data_ptr(), sil_b.data_ptr())
    table.synchronize()
    sil_us = 1e6 * (time.perf_counter() - t6) / 20
    next_rows = dict(
        block_silhouettes=dict(us_per_call=round(sil_us, 1), covered_pixels=int((sil_f > 0).sum()),
                               note="vh_render_blocks (SDFRenderer::drawToFrontAndBack): exact ray/box test of every "
                                    "allocated block's cube, 4 launches, host-timed"),
        icp_round=dict(us_per_round=round(icp_us, 2), pairs=icp_sys[3], algorithmic_bytes=icp_bytes,
                       note="vh_icp_build_system against a raycast target, host-timed and synchronous "
                            "(the step API returns each round's 27 sums to the host)"),
        icp_align=dict(us_per_align=round(align_us, 1), rounds=trk.iterations, us_per_round=round(align_us / max(1, trk.iterations), 2),
                       note="vh_icp_align: all rounds queued at once, 6x6 solve + SE3 update on the device, "
                            "one copy and one synchronisation at the end"),
        preprocess=dict(us_per_frame=round(pre_us, 2), algorithmic_bytes=pre_bytes,
                        achieved_gbs=round(pre_bytes / (pre_us * 1e-6) / 1e9, 1),
                        note="vh_preprocess, back-to-back calls timed on the host (launch gaps included)"),
        garbage_collect=dict(us_per_call=round(1e3 * kt_g["gc_ms"] / max(1, kt_g["gc_calls"]), 2),
                             blocks_freed=gc_counters["last_freed"],
                             note="vh_garbage_collect(0): identify + sweep + release + finish, "
                                  "every block of the last frame freed (8 KiB of voxel traffic each)"))

    # ---- CPU baseline: the oracle on this box's host cores, bounded sample ----
    cpu = None
    if not args.no_cpu_baseline:
        import oracle as O
        op = O.default_params(numBuckets=wl["buckets"], numVoxelBlocks=wl["blocks"], voxelSize=wl["voxel"])
        ot = O.OracleTable(op, Wd, Ht, O.SEM_PINHOLE)
        # the same frame sequence on the host: a few seconds on one thread, then the rest of the budget
        # on `threads` threads (OpenMP; identical results, tests/test_oracle_anchors.py)
        budget_s, nmax = 15.0, args.cpu_frames or 10 ** 9
        done, spent, one_done, one_spent = 0, 0.0, 0, 0.0
        while one_done < nmax and one_spent < 3.0 and one_done < 5000:
            k = one_done % nframes
            v = verts[k].cpu().numpy()
            c0 = time.perf_counter()
            ot.integrate(poses[k], v)
            one_spent += time.perf_counter() - c0
            one_done += 1
        # thread count: the fastest of a short calibration (on the pool's 2 x 64-core hosts the rate
        # peaks at 16 threads -- 245 frames/s -- and falls off beyond; the container's CPU share
        # is not the 256 logical cores it sees)
        threads, best = 1, one_done / one_spent
        cal = []
        for th in (4, 8, 16, 32, 64):
            if th > (os.cpu_count() or 1):
                break
            c0, n = time.perf_counter(), 0
            while time.perf_counter() - c0 < 0.4:
                k = (one_done + n) % nframes
                ot.integrate_mt(poses[k], verts[k].cpu().numpy(), th)
                n += 1
            rate = n / (time.perf_counter() - c0)         # includes the device->host copy: only a ranking
            cal.append((th, round(rate, 1)))
            if rate > best:
                threads, best = th, rate
        while threads > 1 and done < nmax and (args.cpu_frames or spent < budget_s - one_spent) and done < 20000:
            k = (one_done + done) % nframes
            v = verts[k].cpu().numpy()
            c0 = time.perf_counter()
            ot.integrate_mt(poses[k], v, threads)
            spent += time.perf_counter() - c0
            done += 1
        if threads == 1:
            done, spent = one_done, one_spent
        cpu = dict(value=round(done / spent, 3), unit="frames/s", cores=threads, kind="port",
                   one_thread_frames_per_s=round(one_done / one_spent, 3),
                   thread_calibration=cal,
                   sample=f"{one_done} frames on 1 thread, then {done} frames on {threads} threads (fastest of the "
                          f"calibration) of the same {args.workload} sequence, oracle/vh_oracle.c "
                          f"(gcc -O2 -ffp-contract=off -fopenmp), {os.cpu_count()} logical host cores")
        ot.close()

    out = dict(
        metric=baseline_metric(Wd, Ht),
        value=round(fps, 1), unit="frames/s", n_gpus=1, steps=args.steps, warmup=args.warmup,
        ms_per_step=round(1e3 * elapsed / args.steps, 5), higher_is_better=True, scaling="weak",
        vs_baseline=None, dtype="f32", data="synthetic",
        metric_note="value = frames/s TSDF-integrated; the raycast half of the metric is raycast_mpix_per_s",
        config=dict(workload=wl["desc"], resident_frames=nframes, semantics="pinhole",
                    occupied_blocks=occ, allocated_blocks=counters["allocated_total"],
                    keys_last_frame=keys),
        roofline=roofline, cpu_baseline=cpu,
        raycast_mpix_per_s=round(raycast_mpix, 1) if raycast_mpix else None,
        raycast=raycast, next_rows=next_rows, sensor_depth_input=sensor_depth,
        occupancy_index_variant=index_variant,
        kernels=kernels_us,
        frame_algorithmic_bytes=b_frame, frame_algorithmic_gbs=round(frame_gbs, 1),
        frame_frac_of_hbm_peak=round(frame_gbs / HBM_PEAK_GBS, 4),
    )
    print(json.dumps(out), flush=True)
    trk.close()            # contexts go while the torch stream they were bound to is still alive
    table.close()
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
