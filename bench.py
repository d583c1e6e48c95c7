#!/usr/bin/env python3
"""bench.py -- frames/s of the voxel-hashing TSDF path on MI355X (BASELINE.json metric).

A step = one depth frame taken through SDF_Hashtable::integrate (lock epoch -> allocBlocks ->
flattenIntoBuffer -> integrateDepthMap) on the synthetic room of config C2 (640x480); vertex maps
and poses are resident in HBM before the timed region.  One JSON line on stdout (rank 0).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C2|C3|C5] [--legs a,b,...]

Timing: after a fixed run-in lap and W warm-up steps, windows of EXACTLY K steps are timed, each
bracketed by a barrier and a device synchronisation on both sides; windows are repeated until at
least one second (comparison legs: MIN_TIMED_S) has been timed and `value` is K * frames_per_step /
(median window).  Everything else
in the line (roofline, first lap, C3 sub-record, loaded integrate, raycast, sharded path with one
rank, CPU baseline) is measured after that, outside the timed windows.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import math
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the pool's host driver only supports dmabuf IPC (RCCL / cross-process tensor sharing)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MIN_TIMED_S = 0.3            # windows of K steps are repeated until this much has been timed
L3_BYTES = 256 << 20         # Infinity Cache (MI355X_MICROARCH.md)

WORKLOADS = {
    # BASELINE.json configs[1]
    "C2": dict(width=640, height=480, frames=500, buckets=1 << 20, blocks=1 << 18, voxel=0.02, loop=500,
               desc="C2: synthetic 6x3x5 m room, 640x480 x 500-pose camera loop, 2^20 buckets x 5, "
                    "2^18 voxel blocks, voxel 0.02 m, PINHOLE semantics"),
    # C2 with truncation-band allocation (+-10 cm along the viewing ray): hundreds of blocks updated per frame
    "C2band": dict(width=640, height=480, frames=500, buckets=1 << 20, blocks=1 << 18, voxel=0.02, loop=500, band=0.1,
                   options=["band_mode=2"],
                   desc="C2 with vh_set_alloc_band(0.1), band_mode VH_BAND_RAY_DDA: every pixel demands the blocks its viewing "
                        "ray crosses within +-10 cm of its surface point (block DDA), commit + integrateDepthMap under load"),
    # ... the same band as rounds 1-3 specified it: five samples on the ray at half-block steps (reach +-16 cm)
    "C2bandSamples": dict(width=640, height=480, frames=500, buckets=1 << 20, blocks=1 << 18, voxel=0.02, loop=500, band=0.1,
                          desc="C2 with vh_set_alloc_band(0.1), band_mode VH_BAND_RAY: five samples per pixel on the viewing ray"),
    # BASELINE.json configs[2] (HBM-bound stress)
    "C3": dict(width=1280, height=960, frames=2000, buckets=1 << 22, blocks=1 << 21, voxel=0.005, loop=2000, sensor=True,
               desc="C3: synthetic room, 1280x960 x the whole 2000-pose path, 2^22 buckets x 5 (419 MB of VoxelEntry: beyond the "
                    "256 MiB Infinity Cache), 2^21 voxel blocks, voxel 0.005 m, PINHOLE semantics; the 2000 frames resident as "
                    "uint16 sensor images (4.9 GB), fused by vh_integrate_depth_batch"),
    # C3's frames into a table of C5's size on ONE GPU (2^24 buckets x 5 = 1.68 GB of VoxelEntry, 6.5 x the Infinity Cache): the
    # walk with no residency left to argue about ("prove the HBM figure")
    "C5table": dict(width=1280, height=960, frames=64, buckets=1 << 24, blocks=1 << 21, voxel=0.005, loop=2000, sensor=True,
                    desc="C5table: C3's 1280x960 frames into an unsharded table of C5's size, 2^24 buckets x 5 (1.68 GB of "
                         "VoxelEntry = 6.5 x the 256 MiB Infinity Cache), 2^21 voxel blocks, voxel 0.005 m, PINHOLE semantics"),
    # ... and into one of half that size (839 MB): where the claim tiles' placement rule changes over (vh_api_frame.hip: claim_span)
    "C4table": dict(width=1280, height=960, frames=64, buckets=1 << 23, blocks=1 << 21, voxel=0.005, loop=2000, sensor=True,
                    desc="C4table: C3's 1280x960 frames into an unsharded table of 2^23 buckets x 5 (839 MB of VoxelEntry), 2^21 voxel "
                         "blocks, voxel 0.005 m, PINHOLE semantics"),
    # BASELINE.json configs[4], per-rank share when launched with --gpus 8 (one stream per GPU)
    "C5": dict(width=1920, height=1080, frames=64, buckets=1 << 24, blocks=1 << 21, voxel=0.01, loop=500,
               desc="C5: synthetic room, 1920x1080 streams, 2^24 buckets x 5 in all, 2^21 voxel blocks per rank, "
                    "voxel 0.01 m, PINHOLE semantics"),
}
ALL_LEGS = ("two_launch", "first_lap", "index", "sensor", "raycast", "next", "loop", "loaded", "c3", "c5table", "sharded", "cpu")


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default="C2", choices=sorted(WORKLOADS))
    ap.add_argument("--frames", type=int, default=0, help="distinct resident frames (default: workload's)")
    ap.add_argument("--legs", default="all",
                    help="comma list of the extra legs to run after the timed windows: " + ",".join(ALL_LEGS) +
                         " (all / none); the timed windows and the roofline of the dominant kernel always run")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=0, help="frames in the CPU sample (0 = auto, about 15 s)")
    ap.add_argument("--raycast-steps", type=int, default=50)
    ap.add_argument("--profile-steps", type=int, default=200)
    ap.add_argument("--batch", type=int, default=8,
                    help="frames per step: handed to one vh_integrate_batch (one GPU) / carried per camera by one "
                         "all-to-all + all-gather (sharded path)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="sharded path: do not overlap key generation + RCCL with the table work")
    ap.add_argument("--python-exchange", action="store_true",
                    help="sharded path: round 2's Python host (torch.distributed collectives) instead of vh_dist_* inside the library")
    ap.add_argument("--float-packets", action="store_true",
                    help="sharded path: float vertex maps and float camera-z packets instead of uint16 sensor depth")
    ap.add_argument("--sharded-raycast", action="store_true",
                    help="sharded path: also time the raycast over the shards (always on with one rank)")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="vh_set_option on the measured table (flatten_variant=4, walk_nt=0 ...); repeatable")
    ap.add_argument("--sharded", action="store_true",
                    help="force the bucket-range-sharded path (torch.distributed) even with one rank")
    a = ap.parse_args()
    legs = set(ALL_LEGS) if a.legs == "all" else set() if a.legs == "none" else set(a.legs.split(","))
    if legs - set(ALL_LEGS):
        ap.error("--legs: unknown leg(s) " + ",".join(sorted(legs - set(ALL_LEGS))) + "; known: " + ",".join(ALL_LEGS))
    if a.no_cpu_baseline:
        legs.discard("cpu")
    if a.raycast_steps <= 0:
        legs.discard("raycast")
    a.leg_set = legs
    return a


def baseline_metric(width, height):
    """BASELINE.json's metric string for the 640x480 workloads (value = the frames/s half; the raycast
    Mpix/s half travels in `raycast_mpix_per_s`), a plain description otherwise."""
    if (width, height) == (640, 480):
        try:
            return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
        except Exception:
            return "frames/s TSDF-integrated + raycast Mpix/s, 640×480, 1/2/4/8 MI355X"
    return f"frames/s TSDF-integrated + raycast Mpix/s, {width}x{height}"


def timed_windows(step, sync, steps, warmup, first=0, min_time=MIN_TIMED_S, max_windows=4000):
    """W untimed warm-up steps, then windows of exactly `steps` steps (sync on both sides) until
    `min_time` seconds have been timed.  Returns (window times, next step index)."""
    i = first
    for _ in range(warmup):
        step(i)
        i += 1
    sync()
    times = []
    while not times or (sum(times) < min_time and len(times) < max_windows):
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(i)
            i += 1
        sync()
        times.append(time.perf_counter() - t0)
    return times, i


def window_stats(times, steps, frames_per_step=1):
    med = statistics.median(times)
    return dict(value=round(steps * frames_per_step / med, 1), ms_per_step=round(1e3 * med / steps, 5),
                frames_per_step=frames_per_step, windows=len(times),
                timed_s=round(sum(times), 4), window_min_ms=round(1e3 * min(times), 4),
                window_max_ms=round(1e3 * max(times), 4))


def _profiled_kernel(workload, kernel):
    """The instantiation of `kernel` that the committed rocprofv3 --kernel-trace --stats summary of this workload saw most
    often (profiles/kernel_stats_latest.json) -- a sensor-fed workload launches three builds of the pipelined kernel (first
    launch, steady state, flush): the steady state is the one the line is about.  (name, avg_us) or (None, None)."""
    try:
        ks = json.load(open(os.path.join(ROOT, "profiles", "kernel_stats_latest.json"))).get(workload, {})
    except Exception:
        return None, None
    best = None
    for k, v in ks.items():
        if k.startswith(kernel) and isinstance(v, dict) and (best is None or v.get("calls", 0) > best[1].get("calls", 0)):
            best = (k, v)
    return (best[0], best[1].get("avg_us")) if best else (None, None)


def pmc_traffic(workload, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (profiles/pmc_latest.json:
    FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, plus WRITE_SIZE), or None."""
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json"))).get(workload, {})
    except Exception:
        return None
    name, _ = _profiled_kernel(workload, kernel)
    if name and (name + "_hbm_bytes_per_launch") in pmc:
        return pmc[name + "_hbm_bytes_per_launch"]
    for k, v in pmc.items():
        if k.endswith("_hbm_bytes_per_launch") and k.startswith(kernel):
            return v
    return None


def rocprof_mean_us(workload, kernel):
    """Average dispatch duration of `kernel` in the committed `rocprofv3 --kernel-trace --stats` summary of this workload
    (profiles/kernel_stats_latest.json, written by tools/make_pmc_latest.py from the round's kernel_stats CSV), or None."""
    return _profiled_kernel(workload, kernel)[1]


def render_frames(synth, wl, nframes, dev, torch, sensor=None):
    """Resident frames: float4 vertex maps [n, H, W, 4], or (sensor: the workload's "sensor" flag) uint16 depth images
    [n, H, W] as a sensor delivers them (5000 units = 1 m, Application.cpp:38-42) -- 2 instead of 16 bytes per pixel, so that
    C3's whole 2000-pose path is resident (4.9 GB) instead of its first 200 poses."""
    sensor = wl.get("sensor", False) if sensor is None else sensor
    poses = synth.camera_loop(wl["loop"])[:nframes]
    prims = synth.room_primitives()
    if sensor:
        frames = torch.empty((nframes, wl["height"], wl["width"]), dtype=torch.uint16, device=dev)
        for i in range(nframes):
            v = synth.render_room_verts(poses[i], wl["width"], wl["height"], prims, device=dev)
            frames[i] = (v[:, :, 2] * 5000.0).round().clamp(0, 65535).to(torch.uint16)
    else:
        frames = torch.empty((nframes, wl["height"], wl["width"], 4), dtype=torch.float32, device=dev)
        for i in range(nframes):
            frames[i] = synth.render_room_verts(poses[i], wl["width"], wl["height"], prims, device=dev)
    torch.cuda.synchronize()
    return poses, frames


class Integrator:
    """One table + resident frames; step(i) = vh_integrate of frame i mod nframes."""

    def __init__(self, V, L, wl, poses, verts, local_rank, stream, pipeline=True, batch=8):
        self.V, self.L, self.wl, self.stream, self.pipeline, self.batch = V, L, wl, stream, pipeline, max(1, batch)
        self.nframes = len(poses)
        params = V.default_params(numBuckets=wl["buckets"], numVoxelBlocks=wl["blocks"], voxelSize=wl["voxel"])
        self.table = V.SDFHashtable(params, wl["width"], wl["height"], V.SEM_PINHOLE, device=local_rank, stream=stream)
        if wl.get("band"):
            self.table.set_alloc_band(wl["band"])
        # `value` is the reference's frame: flattenKernel's walk over every VoxelEntry (20*N bytes), which the
        # library no longer runs by default (flatten_variant 4, the walk-free frame: the `walk_free` leg); --option overrides
        self.table.set_option("flatten_variant", 3)
        self.indexed = False
        for kv in wl.get("options", []):
            k, v = kv.split("=")
            self.table.set_option(k, int(v))
            if k == "flatten_variant":
                self.indexed = int(v) == 4
        # pipelined frames: one launch per frame (the commit + TSDF update of frame i ride in the launch of
        # frame i+1); every synchronisation flushes, so a timed window contains all of its frames' work
        self.table.set_option("pipeline", 1 if pipeline else 0)
        self.lib, self.h = self.table._lib, self.table._h
        self.pose_keep = [np.ascontiguousarray(p.reshape(16)) for p in poses]
        self.pose_ptrs = [p.ctypes.data_as(C.POINTER(C.c_float)) for p in self.pose_keep]
        self.vert_ptrs = [verts[i].data_ptr() for i in range(self.nframes)]
        # uint16 sensor images instead of vertex maps: vh_integrate_depth(_batch), the vertices computed inside the claim phase
        self.sensor = str(verts.dtype) == "torch.uint16"
        if self.sensor:
            from voxelhashing_demo_amd import synth as _synth
            k_inv = np.linalg.inv(_synth.K_matrix(wl["width"], wl["height"]).astype(np.float64)).astype(np.float32)
            self._kin = np.ascontiguousarray(k_inv.reshape(9))
            self.kin_p = self._kin.ctypes.data_as(C.POINTER(C.c_float))
        # a step = one batch of `batch` consecutive frames handed to vh_integrate_batch (as the sharded path's
        # step is one exchange of `batch` frames per camera): argument blocks prepared once per start frame
        B, n = self.batch, self.nframes
        self._batch_poses = [np.ascontiguousarray(np.stack([self.pose_keep[(k + j) % n] for j in range(B)])) for k in range(n)]
        self._batch_pose_ptrs = [a.ctypes.data_as(C.POINTER(C.c_float)) for a in self._batch_poses]
        self._batch_verts = [(C.c_void_p * B)(*[self.vert_ptrs[(k + j) % n] for j in range(B)]) for k in range(n)]

    def step(self, i):
        k = (i * self.batch) % self.nframes
        if not self.pipeline:                     # the same frames one vh_integrate (two launches) at a time
            for j in range(self.batch):
                self.frame(k + j)
            return
        if self.sensor:
            rc = self.lib.vh_integrate_depth_batch(self.h, self.batch, self._batch_pose_ptrs[k], self._batch_verts[k], self.kin_p)
        else:
            rc = self.lib.vh_integrate_batch(self.h, self.batch, self._batch_pose_ptrs[k], self._batch_verts[k], None)
        if rc != 0:
            self.L.check(rc, "vh_integrate_batch")

    def frame(self, i):
        """one frame (the legs that are not batch-shaped)"""
        k = i % self.nframes
        if self.sensor:
            rc = self.lib.vh_integrate_depth(self.h, self.pose_ptrs[k], self.vert_ptrs[k], self.kin_p)
        else:
            rc = self.lib.vh_integrate(self.h, self.pose_ptrs[k], self.vert_ptrs[k], None)
        if rc != 0:
            self.L.check(rc, "vh_integrate")

    def sync(self):
        self.table.synchronize()

    def mean_occupied(self, samples=125):
        """Mean visible-block count (vh_counters.occupied) over `samples` evenly spaced resident frames, each fused once more and
        its counter read behind a synchronisation (untimed)."""
        n = min(samples, self.nframes)
        total = 0
        for i in range(n):
            self.frame((i * self.nframes) // n)
            total += self.table.counters()["occupied"]
        return int(round(total / max(1, n)))

    def kernel_profile(self, n, first):
        """Per-dispatch HIP events on the path's own stream (untimed pass)."""
        self.table.set_profiling(True)
        for i in range(n):
            self.step(first + i)
        kt = self.table.kernel_times(reset=True)
        self.table.set_profiling(False)
        return kt

    def dominant_roofline(self, workload, kt, occ):
        wl = self.wl
        Wd, Ht, n_entries = wl["width"], wl["height"], self.table.num_entries
        launches = max(1, kt["launches"])
        table_bytes = 20 * n_entries
        resident = table_bytes + 16 * Wd * Ht + 8212 * occ < L3_BYTES
        residency = (("the %.0f MB table plus the frame's stream fit the 256 MiB Infinity Cache: in steady state the walk "
                      "is served on-die, so this is a fraction of the HBM PEAK, not measured HBM traffic (FETCH_SIZE counts "
                      "Infinity-Cache hits); the HBM-resident figure is configs.C3.roofline" % (table_bytes / 1e6))
                     if resident else
                     ("the %.0f MB table exceeds the 256 MiB Infinity Cache: streamed from HBM every frame" % (table_bytes / 1e6)))
        self.commit_roofline = None
        if kt.get("frame_pipelined_ms", 0) > 0:
            # the ONE launch of a pipelined frame: {claim || walk} of frame i+1 and {commit + TSDF update} of frame
            # i.  Its algorithmic bytes are SURVEY.md 8(d)'s B_frame: the vertex map read once (16*W*H), the
            # depth the update gathers (4*W*H), one pass over the table (20*N), the compact entries written
            # (20*occ), per visible block its entry and 4 KiB read + 4 KiB written, one 100-byte bucket probe
            # per distinct block key (keys ~ occ).
            kname = "frame_pipelined_kernel"
            us = 1e3 * kt["frame_pipelined_ms"] / launches
            nbytes = self.input_bytes() + 20 * n_entries + 20 * occ + occ * (20 + 4096 + 4096) + 100 * occ
            if self.indexed:      # (--option flatten_variant=4 on the measured table, the profile runs of the walk-free frame: its own byte count)
                c = self.table.counters()
                nbytes += wl["buckets"] // 8 + 100 * (c["allocated_total"] - c.get("freed_total", 0)) - 20 * n_entries
        else:
            # dominant kernel of the two-launch frame: per-pixel claim phase || walk over the VoxelEntry array:
            # the vertex map read once by the claim half (16*W*H), one pass over the table (20*N), the compact
            # entries written (20*occ) and one 100-byte bucket probe per distinct block key (keys ~ occ).
            kname = "frame_scan_claim_kernel"
            us = 1e3 * kt["frame_scan_claim_ms"] / launches
            nbytes = (2 if self.sensor else 16) * Wd * Ht + 20 * n_entries + 20 * occ + 100 * occ
            # launch 2: per occupied block the 20-byte entry, 4 KiB of voxels read and 4 KiB written, plus the
            # depth plane the update gathers from (counted once)
            us2 = 1e3 * kt["frame_commit_integrate_ms"] / launches
            bytes2 = occ * (20 + 4096 + 4096) + (2 if self.sensor else 4) * Wd * Ht
            ach2 = bytes2 / (us2 * 1e-6) / 1e9 if us2 > 0 else 0.0
            self.commit_roofline = dict(bound="hbm", kernel="frame_commit_integrate_kernel", achieved=round(ach2, 1),
                                        peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach2 / HBM_PEAK_GBS, 4),
                                        traffic=pmc_traffic(workload, "frame_commit_integrate_kernel"),
                                        bytes_per_launch=bytes2, us_per_launch=round(us2, 2))
        achieved = nbytes / (us * 1e-6) / 1e9 if us > 0 else 0.0
        # SURVEY.md 8(d) "HBM-read roofline": the frame's bytes without its write terms (compact entries and voxels written)
        read_bytes = nbytes - 20 * occ - (4096 * occ if kname == "frame_pipelined_kernel" else 0)
        out = dict(bound="hbm", kernel=kname, achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                   frac=round(achieved / HBM_PEAK_GBS, 4), traffic=pmc_traffic(workload, kname),
                   bytes_per_launch=nbytes, us_per_launch=round(us, 2),
                   hbm_read_frac=round(read_bytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if us > 0 else None,
                   residency=residency)
        # the same fraction on the average duration rocprofv3 reported for this kernel in the committed profile of this
        # workload (another box, the whole run's dispatches): the figure a reader recomputes from profiles/
        rp = rocprof_mean_us(workload, kname)
        if rp:
            out["rocprofv3_us_per_launch"] = rp
            out["frac_at_rocprofv3_mean"] = round(nbytes / (rp * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
        return out

    def input_bytes(self):
        """The frame's input terms of SURVEY.md 8(d): the vertex map read once (16*W*H) + the depth the TSDF update gathers
        (4*W*H) -- or, fed by the sensor image: the image read once (2*W*H) + the depth gathered from its copy (2*W*H)."""
        Wd, Ht = self.wl["width"], self.wl["height"]
        return (2 + 2) * Wd * Ht if self.sensor else (16 + 4) * Wd * Ht

    def close(self):
        self.table.close()


def measure_workload(args, V, L, synth, torch, name, local_rank, steps, warmup, want_profile=True, frames=None,
                     pipeline=True):
    """Timed windows + dominant-kernel roofline of one single-GPU workload.  Returns (record, Integrator, poses, verts)."""
    wl = dict(WORKLOADS[name], options=list(WORKLOADS[name].get("options", [])) + list(getattr(args, "option", [])))
    dev = torch.device("cuda", local_rank)
    stream = torch.cuda.Stream(device=dev)
    nframes = args.frames or wl["frames"]
    poses, verts = frames if frames is not None else render_frames(synth, wl, nframes, dev, torch)
    nframes = len(poses)
    it = Integrator(V, L, wl, poses, verts, local_rank, stream, pipeline=pipeline, batch=args.batch)
    B = it.batch
    lap = -(-(max(nframes, 500) if name.startswith("C2") else nframes) // B)
    for i in range(lap):           # fixed, untimed run-in (one lap of the resident frames) before the W warm-up steps
        it.step(i)
    it.sync()
    # (the headline windows are repeated until a full second has been timed; the comparison legs settle for 0.3 s)
    times, nxt = timed_windows(it.step, lambda: (it.sync(), torch.cuda.synchronize()), steps, warmup, first=lap,
                               min_time=1.0)
    rec = window_stats(times, steps, B)
    counters = it.table.counters()
    if counters["spin_timeouts"]:          # (a serialised launch gave up waiting: the frames timed are not the frames fused)
        raise SystemExit(f"bench.py: {counters['spin_timeouts']} workgroups of serialised launches timed out (vh_counters.spin_timeouts)")
    rec["occupied_blocks_last_frame"], rec["allocated_blocks"], rec["resident_frames"] = counters["occupied"], counters["allocated_total"], nframes
    # the visible-block count the byte formulas use is the MEAN over the resident frames (the counter passes and the per-dispatch
    # times are means over all of them too; the last frame's count alone made the walk-free frame's traffic look 1.4 x its bytes)
    occ = it.mean_occupied()
    rec["occupied_blocks"] = occ
    if want_profile and args.profile_steps > 0:
        kt = it.kernel_profile(args.profile_steps, nxt)
        rec["roofline"] = it.dominant_roofline(name, kt, occ)
        if it.commit_roofline:
            rec["roofline_commit_integrate"] = it.commit_roofline
        rec["kernels"] = {k[:-3] + "_us": round(1e3 * v / max(1, kt["launches"]), 2)
                          for k, v in kt.items() if k.endswith("_ms") and v > 0 and k not in ("raycast_ms",)}
    Wd, Ht, n_entries = wl["width"], wl["height"], it.table.num_entries
    # algorithmic bytes of the whole frame (SURVEY.md 8(d) B_frame; no mutex memset in this build);
    # distinct in-frustum block keys of a frame ~ occupied blocks (not counted on the device)
    b_frame = it.input_bytes() + 20 * n_entries + 20 * occ + occ * (20 + 4096 + 4096) + 100 * occ
    rec["frame_algorithmic_bytes"] = b_frame
    rec["input"] = ("uint16 sensor images (vh_integrate_depth_batch): input terms 2*W*H + 2*W*H in place of 16*W*H + 4*W*H"
                    if it.sensor else "float4 vertex maps (vh_integrate_batch)")
    rec["ms_per_frame"] = round(rec["ms_per_step"] / B, 5)
    rec["frame_algorithmic_gbs"] = round(b_frame * rec["value"] / 1e9, 1)
    rec["frame_frac_of_hbm_peak"] = round(b_frame * rec["value"] / 1e9 / HBM_PEAK_GBS, 4)
    return rec, it, poses, verts


def two_launch_record(args, it, name, occ, steps, warmup, sync):
    """The same frames unpipelined: vh_integrate as two launches per frame ({claim || walk}, {commit + TSDF
    update}); what round 1 measured, and what a caller gets who reads the model between frames."""
    it.table.set_option("pipeline", 0)
    it.pipeline = False
    t_2, nxt2 = timed_windows(it.step, sync, steps, warmup)
    rec2 = dict(window_stats(t_2, steps, it.batch), unit="frames/s")
    if args.profile_steps > 0:
        kt2 = it.kernel_profile(args.profile_steps, nxt2)
        rec2["roofline_scan_claim"] = it.dominant_roofline(name, kt2, occ)
        rec2["roofline_commit_integrate"] = it.commit_roofline
        rec2["kernels"] = {k[:-3] + "_us": round(1e3 * v / max(1, kt2["launches"]), 2)
                           for k, v in kt2.items() if k.endswith("_ms") and v > 0 and k not in ("raycast_ms",)}
    it.table.set_option("pipeline", 1)
    it.pipeline = True
    return rec2


def index_variant_record(args, it, name, steps, warmup, sync):
    """The walk-free frame (vh_set_option flatten_variant = 4) on the same resident frames, pipelined like the headline
    path: the walk role reads the bucket-occupancy bitmap and the non-empty buckets instead of every VoxelEntry.  NOT the
    reference's flattenKernel (SURVEY.md 8(d) allows it if the smaller byte count is reported), hence never `value`.  Its
    roofline is on that smaller byte count; the launch streams next to nothing, so what bounds it is named as measured:
    chains of dependent reads and VALU issue of the claim tiles and the TSDF update (profiles/r05_index_roles.txt)."""
    wl = it.wl
    Wd, Ht = wl["width"], wl["height"]
    it.table.set_option("flatten_variant", 4)
    t_idx, nxt = timed_windows(it.step, sync, steps, warmup)
    rec = dict(window_stats(t_idx, steps, it.batch), unit="frames/s")
    counters = it.table.counters()
    occ, alloc = it.mean_occupied(), counters["allocated_total"] - counters.get("freed_total", 0)
    flatten_bytes = wl["buckets"] // 8 + 100 * alloc
    nbytes = it.input_bytes() + flatten_bytes + 20 * occ + occ * (20 + 4096 + 4096) + 100 * occ
    rec["flatten_bytes"] = flatten_bytes
    if args.profile_steps > 0:
        kt = it.kernel_profile(min(args.profile_steps, 200), nxt)
        us = 1e3 * kt["frame_pipelined_ms"] / max(1, kt["launches"])
        ach = nbytes / (us * 1e-6) / 1e9 if us > 0 else 0.0
        rec["roofline"] = dict(
            bound="latency + VALU issue (chains of dependent reads of the claim tiles, the index walk and the TSDF update; "
                  "no stream: the HBM fraction below says how far from a byte bound the launch is, not how good it is)",
            kernel="frame_pipelined_kernel (walk role = flatten_index_tile)", achieved=round(ach, 1), peak=HBM_PEAK_GBS,
            unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4), traffic=pmc_traffic(name + "index", "frame_pipelined_kernel"),
            bytes_per_launch=nbytes, us_per_launch=round(us, 2),
            bytes_formula="16*W*H + 4*W*H + numBuckets/8 + 100*allocated + 20*occ + occ*(20+4096+4096) + 100*occ "
                          "(SURVEY.md 8(d) with numBuckets/8 + 100*allocated in place of 20*N)",
            roles="profiles/r05_index_roles.txt (diagnostics build: each role switched off)")
        rp = rocprof_mean_us(name + "index", "frame_pipelined_kernel")
        if rp:
            rec["roofline"]["rocprofv3_us_per_launch"] = rp
            rec["roofline"]["frac_at_rocprofv3_mean"] = round(nbytes / (rp * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
    it.table.set_option("flatten_variant", 3)
    # What bounds this launch is not bytes: a latency / issue model beside the measured time.  Constants measured
    # on C2 with roles switched off in a diagnostics build (profiles/r06_index_roles_C2.txt): an empty launch of this grid (dispatch
    # + the commit role's ticket chain) 4.5 us; the claim tiles alone 7.3 us = 4 dependent round trips (vertex -> bucket's first slot
    # -> pending claim word -> atomicMax) of ~0.7 us on top of it.  Vector issue: ~300 wave-instructions per claim wave of 64 pixels
    # (IEEE divisions of world2Voxel, frustum test, hash, dedup) and 912 per updated block (SQ_INSTS_VALU of the TSDF update on C3,
    # profiles/r05_pmc_commit_integrate_C3_*.json), 4 cycles each on 1 024 SIMDs at 2.4 GHz.  model = floor + the longer of the two.
    floor_us, round_trip_us, claim_trips = 4.5, 0.7, 4
    issue_us = ((Wd * Ht / 64.0) * 300.0 + occ * 912.0) / 1024.0 * 4.0 / 2400.0
    rec["latency_model"] = dict(launch_floor_us=floor_us, claim_chain_us=round(claim_trips * round_trip_us, 2),
                                vector_issue_us=round(issue_us, 2), model_us=round(floor_us + max(claim_trips * round_trip_us, issue_us), 2),
                                source="profiles/r06_index_roles_C2.txt (floor, round trip); 300 / 912 vector wave-instructions per claim wave / "
                                       "updated block at 4 cycles on 1 024 SIMDs, 2.4 GHz")
    rec["note"] = ("vh_set_option(flatten_variant=4), pipelined like the headline path: walk over the bucket-occupancy "
                   "bitmap (numBuckets/8 bytes) + the non-empty buckets instead of the 20*N-byte table walk; NOT the "
                   "reference's flattenKernel, hence not `value`; bit-equal tables, voxels and compact set "
                   "(tests/test_gpu_bench_paths.py, test_gpu_full_size.py)")
    return rec


def cpu_baseline_leg(args, name, poses, verts, budget_s=15.0, one_thread_s=3.0, prefix=0):
    """The oracle (kind "port") on this box's host cores over a bounded sample of the same frame
    sequence: `one_thread_s` seconds on one thread, then the rest of `budget_s` on the fastest thread count.
    prefix > 0: the sample is the first `prefix` frames of the sequence, each leg once ("C2/C3 on a 20-frame
    prefix"), whatever that takes up to the budget."""
    import oracle as O
    wl = WORKLOADS[name]
    Wd, Ht, nframes = wl["width"], wl["height"], len(poses)
    if prefix:
        nframes = min(nframes, prefix)
    host = [np.ascontiguousarray(verts[k].cpu().numpy()) for k in range(nframes)] if nframes <= 64 else None
    frame = (lambda k: host[k]) if host is not None else (lambda k: verts[k].cpu().numpy())
    op = O.default_params(numBuckets=wl["buckets"], numVoxelBlocks=min(wl["blocks"], 1 << 18), voxelSize=wl["voxel"])
    ot = O.OracleTable(op, Wd, Ht, O.SEM_PINHOLE)
    nmax = args.cpu_frames or (prefix if prefix else 10 ** 9)
    done, spent, one_done, one_spent = 0, 0.0, 0, 0.0
    while one_done < nmax and one_spent < one_thread_s and one_done < 5000:
        k = one_done % nframes
        v = frame(k)
        c0 = time.perf_counter()
        ot.integrate(poses[k], v)
        one_spent += time.perf_counter() - c0
        one_done += 1
    # thread count: the fastest of a short calibration (on the pool's 2 x 64-core hosts the rate
    # peaks at 16 threads and falls off beyond; the container's CPU share is not the 256 logical
    # cores it sees)
    threads, best, cal = 1, one_done / one_spent, []
    per_frame = one_spent / one_done
    for th in (4, 8, 16, 32, 64):
        if th > (os.cpu_count() or 1):
            break
        c0, n = time.perf_counter(), 0
        while time.perf_counter() - c0 < max(0.4, 1.5 * per_frame / th):
            k = (one_done + n) % nframes
            ot.integrate_mt(poses[k], frame(k), th)
            n += 1
        rate = n / (time.perf_counter() - c0)         # includes the host copy where frames are not staged: only a ranking
        cal.append((th, round(rate, 1)))
        if rate > best:
            threads, best = th, rate
    while threads > 1 and done < nmax and (args.cpu_frames or spent < budget_s - one_spent) and done < 20000:
        k = (one_done + done) % nframes
        v = frame(k)
        c0 = time.perf_counter()
        ot.integrate_mt(poses[k], v, threads)
        spent += time.perf_counter() - c0
        done += 1
    if threads == 1:
        done, spent = one_done, one_spent
    ot.close()
    return dict(value=round(done / spent, 3), unit="frames/s", cores=threads, kind="port",
                one_thread_frames_per_s=round(one_done / one_spent, 3), thread_calibration=cal,
                sample_short=f"{one_done} fr x1 thread + {done} fr x{threads} threads, {name}"
                             + (f" {nframes}-frame prefix" if prefix else "") + f", {os.cpu_count()} logical cores",
                sample=f"{one_done} frames on 1 thread, then {done} frames on {threads} threads (fastest of the "
                       f"calibration) of the same {name} sequence" + (f" (its first {nframes} frames, cycled)" if prefix else "")
                       + ", oracle/vh_oracle.c "
                       f"(gcc -O2 -ffp-contract=off -fopenmp), {os.cpu_count()} logical host cores")


def cpu_baseline_c1(synth):
    """BASELINE.json configs[0], literally: ONE 640x480 synthetic sphere depth frame, 8^3 blocks, 2^17 buckets, scalar CPU
    integrate -- the oracle on one thread, both sphere scenes of SURVEY.md 8(c) G2/G3 (inside-out R = 2 m, from outside
    r = 0.5 m at 1.5 m), REFERENCE semantics (what the anchors pin) and PINHOLE; a fresh table per repetition, the
    table's allocation outside the clock; median of the repetitions that fit ~0.5 s per case."""
    import oracle as O
    I4 = np.eye(4, dtype=np.float32)
    out = {}
    for scene, verts in (("sphere_inside", synth.sphere_inside_scene()), ("sphere_outside", synth.sphere_outside_scene())):
        for sem, sname in ((O.SEM_REFERENCE, "reference"), (O.SEM_PINHOLE, "pinhole")):
            times, blocks = [], 0
            while sum(times) < 0.5 and len(times) < 50:
                ot = O.OracleTable(O.default_params(numBuckets=1 << 17, numVoxelBlocks=4096), 640, 480, sem)
                c0 = time.perf_counter()
                ot.integrate(I4, verts)
                times.append(time.perf_counter() - c0)
                blocks = ot.compact_count()
                ot.close()
            med = statistics.median(times)
            out[f"{scene}_{sname}"] = dict(ms_per_frame=round(1e3 * med, 3), frames_per_s=round(1.0 / med, 2),
                                           occupied_blocks=blocks, repetitions=len(times))
    head = out["sphere_inside_pinhole"]
    return dict(value=head["frames_per_s"], unit="frames/s", cores=1, kind="port",
                sample_short="C1: 1 sphere frame (inside-out, pinhole), fresh 2^17-bucket table, 1 thread",
                sample="C1 (BASELINE.json configs[0]): one 640x480 sphere frame into a fresh 2^17-bucket table, oracle/vh_oracle.c "
                       "on one thread; value = the inside-out sphere in PINHOLE semantics, `cases` holds both scenes x both semantics",
                cases=out)


def launcher_command(argv, gpus, port):
    """What `python bench.py --gpus N` starts when it was not itself started by a launcher: the driver's own command line,
    one rank per GPU of this node (argv = this process's arguments, passed through unchanged).  port 0: the launcher picks a
    free rendezvous port itself (c10d rendezvous on 127.0.0.1:0) instead of one this process found free a moment earlier."""
    head = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}"]
    if port:
        head += ["--master-addr", "127.0.0.1", "--master-port", str(port)]
    else:
        head += ["--rdzv-backend=c10d", "--rdzv-endpoint=127.0.0.1:0", "--local-addr", "127.0.0.1"]
    return head + [os.path.abspath(__file__)] + list(argv)


def self_launch(args, argv):
    """--gpus N > 1 without WORLD_SIZE: start the N ranks as a CHILD process (never exec: this process must not be replaced,
    and it never touches the GPU itself -- no torch import here), relay their output and rank 0's JSON line, return the
    child's exit code.  The child leads a process group of its own; whatever ends this process in an orderly way (SIGTERM
    from a driver's or pytest's timeout, SIGHUP, Ctrl-C) ends the launcher AND its ranks -- terminate, ten seconds of grace,
    kill -- so that no orphaned rank keeps a GPU busy."""
    import signal
    import subprocess
    cmd = launcher_command(argv, args.gpus, 0)
    print("bench.py: starting " + " ".join(cmd), file=sys.stderr, flush=True)
    # HSA_ENABLE_IPC_MODE_LEGACY=0: this pool's host driver only supports dmabuf IPC (without it RCCL's P2P set-up between
    # processes fails with hipIpcGetMemHandle: invalid argument); the image exports it, a caller's own value wins
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "8"))
    def die_with_parent():
        #  a parent that is SIGKILLed -- a harness's hard limit, the OOM killer -- runs no handler: the launcher asks
        # the kernel for SIGTERM at its parent's death (prctl PR_SET_PDEATHSIG), and ends its ranks as it does on any SIGTERM
        try:
            C.CDLL("libc.so.6", use_errno=True).prctl(1, signal.SIGTERM, 0, 0, 0)
        except Exception:
            pass

    child = subprocess.Popen(cmd, env=env, start_new_session=True, preexec_fn=die_with_parent)

    def end_children(signum=None, frame=None):
        for sig, grace in ((signal.SIGTERM, 10.0), (signal.SIGKILL, 5.0)):
            try:
                os.killpg(child.pid, sig)
            except (ProcessLookupError, PermissionError):
                break
            try:
                child.wait(timeout=grace)
                break
            except subprocess.TimeoutExpired:
                continue
        if signum is not None:
            raise SystemExit(128 + signum)

    old = {sig: signal.signal(sig, end_children) for sig in (signal.SIGTERM, signal.SIGHUP)}
    try:
        return child.wait()
    except KeyboardInterrupt:
        end_children()
        return child.wait()
    finally:
        for sig, h in old.items():
            signal.signal(sig, h)


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args, sys.argv[1:]))
    import torch
    import torch.distributed as dist

    import voxelhashing_demo_amd as V
    from voxelhashing_demo_amd import _lib as L
    from voxelhashing_demo_amd import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("VH_BENCH_SHARE_GPU") == "1":          # test rig: all ranks on one device (see init_dist)
        local_rank = 0
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE)")
    if os.environ.get("VH_BENCH_TEST_HOLD_S"):      # test hook (tests/test_bench_launch.py): ranks that stay around to be terminated
        time.sleep(float(os.environ["VH_BENCH_TEST_HOLD_S"]))
    # (device_count() does not initialise the GPU: a rank that cannot have a device of its own says so before any HIP call)
    visible = torch.cuda.device_count()
    if os.environ.get("VH_BENCH_SHARE_GPU") != "1" and visible < world:
        raise SystemExit(f"bench.py: {world} GPUs requested (--gpus {world}), {visible} visible on this node: "
                         "one rank per GPU is needed (RCCL refuses two ranks on one device)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    legs = args.leg_set
    if world > 1 or args.sharded:
        from voxelhashing_demo_amd import dist as vdist
        init_dist(dist, torch, local_rank)
        wl = WORKLOADS[args.workload]
        args.metric_name = baseline_metric(wl["width"], wl["height"])
        out = vdist.bench_sharded(args, wl, args.workload, rank, world, local_rank)
        if rank == 0:
            if "cpu" in legs:       # the reported CPU baseline: one camera's frames of the same workload, rank 0's host
                poses, verts = render_frames(synth, wl, min(32, wl["frames"]), torch.device("cuda", local_rank), torch, sensor=False)
                out["cpu_baseline"] = cpu_baseline_leg(args, args.workload, poses, verts)
        dist.barrier()
        dist.destroy_process_group()
        if rank == 0:
            emit(out, compact_sharded_line(out))
        return

    # A full collection of the Python garbage collector walks every object torch has created (~10^6):
    # 30-50 ms, i.e. longer than a timed window.  Existing objects are moved out of its reach
    # and it stays off while the clock runs (the loops allocate only small short-lived objects).
    import gc
    gc.collect()
    gc.freeze()
    gc.disable()

    name = args.workload
    wl = WORKLOADS[name]
    Wd, Ht = wl["width"], wl["height"]
    main_rec, it, poses, verts = measure_workload(args, V, L, synth, torch, name, local_rank, args.steps, args.warmup)
    table, lib, h, nframes = it.table, it.lib, it.h, it.nframes
    pose_ptrs = it.pose_ptrs
    dev = torch.device("cuda", local_rank)
    stream = it.stream
    occ = main_rec["occupied_blocks"]
    sync = lambda: (table.synchronize(), torch.cuda.synchronize())        # noqa: E731
    extra = {}

    # ---- the same frames unpipelined: vh_integrate as two launches per frame ({claim || walk}, {commit + TSDF
    # update}); what round 1 measured, and what a caller gets who reads the model between frames ----
    if "two_launch" in legs:
        extra["two_launch_frame"] = two_launch_record(args, it, name, occ, args.steps, args.warmup, sync)

    # ---- first lap: a FRESH table over the resident frames in order -- the frames in which blocks are
    # actually inserted (commit phase, heap pops, new-block integration); the steady-state `value`
    # above only revisits known blocks ----
    if "first_lap" in legs:
        fresh = Integrator(V, L, wl, poses, verts, local_rank, stream)
        fresh.sync()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(nframes):
            fresh.frame(i)
        fresh.sync()
        lap_s = time.perf_counter() - t0
        fc = fresh.table.counters()
        extra["first_lap"] = dict(value=round(nframes / lap_s, 1), unit="frames/s", frames=nframes,
                                  ms_per_step=round(1e3 * lap_s / nframes, 5), blocks_inserted=fc["allocated_total"],
                                  note="fresh table, resident frames 0..n-1 once, in order: every block of the model is "
                                       "inserted inside this window")
        fresh.close()

    # ---- same workload with the opt-in occupancy-index walk (NOT the reference algorithm: the
    # flatten step reads the 1-bit-per-bucket index and only the non-empty buckets instead of
    # every VoxelEntry; reported separately, never as `value`) ----
    if "index" in legs:
        extra["occupancy_index_variant"] = index_variant_record(args, it, name, args.steps, args.warmup, sync)

    # ---- the same frames straight from uint16 sensor depth (vh_integrate_depth: preProcess's vertex
    # computation inside the claim phase, no vertex map in memory), against the two-call form
    # vh_preprocess + vh_integrate.  Extension of the boundary; reported separately. ----
    k_inv = np.linalg.inv(synth.K_matrix(Wd, Ht).astype(np.float64)).astype(np.float32)
    kin = np.ascontiguousarray(k_inv.reshape(9))
    kin_p = kin.ctypes.data_as(C.POINTER(C.c_float))
    if "sensor" in legs and not it.sensor:      # (a sensor-fed workload IS this leg's fused form already)
        nd = min(nframes, 250)
        depth16 = torch.empty((nd, Ht, Wd), dtype=torch.uint16, device=dev)
        for i in range(nd):
            depth16[i] = (verts[i, :, :, 2] * 5000.0).round().clamp(0, 65535).to(torch.uint16)
        d_ptrs = [depth16[i].data_ptr() for i in range(nd)]
        tmp_v, tmp_n = torch.empty_like(verts[0]), torch.empty_like(verts[0])
        tv, tn, st = tmp_v.data_ptr(), tmp_n.data_ptr(), C.c_void_p(stream.cuda_stream)
        torch.cuda.synchronize()   # the images were written on torch's default stream, the table reads them on its own

        def fused_depth(i):
            lib.vh_integrate_depth(h, pose_ptrs[i % nd], d_ptrs[i % nd], kin_p)

        def two_calls(i):
            lib.vh_preprocess(d_ptrs[i % nd], kin_p, Wd, Ht, tv, tn, st)
            lib.vh_integrate(h, pose_ptrs[i % nd], tv, None)

        sensor = {}
        for label, fn in (("vh_integrate_depth", fused_depth), ("vh_preprocess + vh_integrate", two_calls)):
            t_s, _ = timed_windows(fn, sync, args.steps * it.batch, args.warmup)
            sensor[label] = window_stats(t_s, args.steps * it.batch)["value"]
        extra["sensor_depth_input"] = dict(
            frames_per_s=sensor, unit="frames/s",
            note="input: uint16 depth images (2 B/pixel); the fused call computes the vertices inside the claim "
                 "phase and gathers depth from the image, bit-equal to the two-call form")
        del depth16

    # ---- raycast Mpix/s (second half of the metric): end to end and per-dispatch over the SAME poses ----
    raycast_mpix = None
    if "raycast" in legs:
        depth = torch.empty((Ht, Wd), dtype=torch.float32, device=dev)
        dptr = depth.data_ptr()
        ray_poses = [(7 * i) % nframes for i in range(args.raycast_steps)]
        torch.cuda.synchronize()
        for k in ray_poses[:5]:
            lib.vh_raycast(h, pose_ptrs[k], 0.1, 5.0, dptr)
        sync()
        t1 = time.perf_counter()
        for k in ray_poses:
            lib.vh_raycast(h, pose_ptrs[k], 0.1, 5.0, dptr)
        sync()
        ray_s = time.perf_counter() - t1
        raycast_mpix = len(ray_poses) * Wd * Ht / ray_s / 1e6
        table.set_profiling(True)
        for k in ray_poses:
            lib.vh_raycast(h, pose_ptrs[k], 0.1, 5.0, dptr)
        kt_r = table.kernel_times(reset=True)
        table.set_profiling(False)
        raycast_us = 1e3 * kt_r["raycast_ms"] / max(1, kt_r["raycast_launches"])
        # SURVEY.md 8(d) raycast work unit: every visible block and the output touched once
        ray_bytes = (4096 + 20) * occ + 4 * Wd * Ht
        rc = dict(mpix_per_s=round(raycast_mpix, 1), kernel_us=round(raycast_us, 2), poses=len(ray_poses),
                  kernel_mpix_per_s=round(Wd * Ht / raycast_us, 1) if raycast_us > 0 else None,
                  algorithmic_bytes=ray_bytes,
                  achieved_gbs=round(ray_bytes / (raycast_us * 1e-6) / 1e9, 1) if raycast_us > 0 else None,
                  note="kernel_us is the mean HIP-event duration over the same poses as the end-to-end loop")
        rc["roofline"] = raycast_roofline(name, raycast_us, Wd, Ht)
        rc["traversal"] = ("voxel DDA (raycastSDF.frag:121-177 re-specified; cooperative form: one block list per 8x8 patch, idle waves of a workgroup take items of its other patches), "
                           "vh_set_option raycast_mode = VH_RAYCAST_DDA, the default")
        # the same poses with the normal map written by the same pass, and with the fixed-step march of rounds 1-2
        normals = torch.empty((Ht, Wd, 4), dtype=torch.float32, device=dev)
        variants = {}
        for label, mode, with_normals in (("dda_with_normals", 1, True), ("fixed_step_march", 0, False)):
            table.set_option("raycast_mode", mode)
            table.set_profiling(True)
            for k in ray_poses:
                if with_normals:
                    lib.vh_raycast_normals(h, pose_ptrs[k], 0.1, 5.0, dptr, normals.data_ptr())
                else:
                    lib.vh_raycast(h, pose_ptrs[k], 0.1, 5.0, dptr)
            kt_v = table.kernel_times(reset=True)
            table.set_profiling(False)
            variants[label] = round(1e3 * kt_v["raycast_ms"] / max(1, kt_v["raycast_launches"]), 2)
        table.set_option("raycast_mode", 1)
        rc["variants_kernel_us"] = variants
        extra["raycast"] = rc

    # ---- loaded integrate: truncation-band allocation on (every pixel demands the blocks within
    # +-10 cm of its surface point), so thousands of 8^3 blocks are updated per frame and launch 2
    # (commit + integrateDepthMap) is measured under load: 20 B entry + 4 KiB read + 4 KiB written per block ----
    if "loaded" in legs and name == "C2":
        l_rec, l_it, _, _ = measure_workload(args, V, L, synth, torch, "C2band", local_rank, args.steps, args.warmup,
                                             frames=(poses, verts))
        l_rec["unit"], l_rec["workload"] = "frames/s", WORKLOADS["C2band"]["desc"]
        extra["loaded_integrate"] = l_rec
        # launch 2 alone under this load (two-launch frames): the 8^3-block read-modify-write by itself
        l_it.table.set_option("pipeline", 0)
        l_it.pipeline = False
        for i in range(20):
            l_it.step(i)
        kt_l2 = l_it.kernel_profile(min(200, max(20, args.profile_steps)), 20)
        l_it.dominant_roofline("C2band", kt_l2, l_it.mean_occupied())
        l_rec["roofline_commit_integrate_two_launch"] = l_it.commit_roofline
        l_it.close()
        # the band as rounds 1-3 specified it (five ray samples per pixel), for comparison
        s_rec, s_it, _, _ = measure_workload(args, V, L, synth, torch, "C2bandSamples", local_rank, max(50, args.steps // 4),
                                             min(args.warmup, 20), frames=(poses, verts))
        l_rec["ray_samples_variant"] = dict(value=s_rec["value"], unit="frames/s", occupied_blocks=s_rec["occupied_blocks"],
                                            roofline=s_rec.get("roofline"), workload=WORKLOADS["C2bandSamples"]["desc"])
        s_it.close()

    # ---- next rows ----
    # (the legs below hand float4 vertex maps to vh_preprocess / ICP / the oracle: a sensor-fed workload renders a few)
    fposes, fverts = (poses, verts) if not it.sensor else render_frames(synth, wl, min(nframes, 24), dev, torch, sensor=False)
    if "next" in legs:
        extra["next_rows"] = next_rows_leg(V, synth, torch, it, fverts, k_inv, stream, Wd, Ht)
    # ---- closed loop: track each frame against the model, integrate at the tracked pose, raycast ----
    if "loop" in legs:
        try:
            extra["closed_loop"] = closed_loop_leg(V, synth, torch, wl, fposes, fverts, local_rank, stream)
        except RuntimeError as e:        # (a view ICP cannot track -- VH_ERR_SINGULAR on a single plane -- must not cost the line)
            extra["closed_loop"] = dict(error=str(e)[:200])

    # ---- C3 sub-record: the table that does NOT fit the Infinity Cache (true HBM streaming) ----
    if "c3" in legs and name == "C2":
        it.close()
        it = None
        del verts
        torch.cuda.empty_cache()
        c3_rec, c3_it, c3_poses, c3_verts = measure_workload(args, V, L, synth, torch, "C3", local_rank,
                                                              max(50, min(args.steps, 200)), min(args.warmup, 20))
        c3_rec["workload"] = WORKLOADS["C3"]["desc"]
        c3_rec["unit"] = "frames/s"
        if "two_launch" in legs:
            c3_rec["two_launch_frame"] = two_launch_record(
                args, c3_it, "C3", c3_rec["occupied_blocks"], max(50, min(args.steps, 200)), min(args.warmup, 20),
                lambda: (c3_it.sync(), torch.cuda.synchronize()))
        if "index" in legs:
            c3_rec["occupancy_index_variant"] = index_variant_record(
                args, c3_it, "C3", max(50, min(args.steps, 200)), min(args.warmup, 20),
                lambda: (c3_it.sync(), torch.cuda.synchronize()))
        if "raycast" in legs:
            # the raycast of the C3 model (1280x960, 5 mm voxels: the per-lane walk behind the beam front end, chosen by the view)
            W3, H3 = WORKLOADS["C3"]["width"], WORKLOADS["C3"]["height"]
            d3 = torch.empty((H3, W3), dtype=torch.float32, device=dev)
            c3_it.table.set_profiling(True)
            stats = {}
            for mode, label in ((V.RAYCAST_DDA, "dda"), (V.RAYCAST_FIXED_STEP, "fixed_step_march")):
                c3_it.table.set_raycast_mode(mode)
                c3_it.table.raycast(c3_poses[0], d3)
                c3_it.table.kernel_times()
                for k in range(0, min(len(c3_poses), 60), 3):
                    c3_it.table.raycast(c3_poses[k], d3)
                c3_it.table.synchronize()
                kt3 = c3_it.table.kernel_times()
                stats[label] = round(1e3 * kt3["raycast_ms"] / max(1, kt3["raycast_launches"]), 2)
            c3_it.table.set_raycast_mode(V.RAYCAST_DDA)
            c3_it.table.set_profiling(False)
            c3_rec["raycast"] = dict(kernel_us=stats["dda"], kernel_mpix_per_s=round(W3 * H3 / stats["dda"], 1), poses=20,
                                     variants_kernel_us={"fixed_step_march": stats["fixed_step_march"]},
                                     traversal="voxel DDA; form chosen by the view (raycast_beam 3): at 5 mm voxels 0.1-5 m is one window of "
                                               "4 x 64 half-block slabs, so the cooperative form runs (lists shared inside the workgroup)")
            del d3
        extra["configs"] = {"C3": c3_rec}
        c3_it.close()
        if "cpu" in legs:
            # C3 on a 20-frame prefix, one thread and the fastest thread count
            p20, v20 = render_frames(synth, WORKLOADS["C3"], 20, dev, torch, sensor=False)
            c3_rec["cpu_baseline"] = cpu_baseline_leg(args, "C3", p20, v20, budget_s=10.0, one_thread_s=3.0, prefix=20)
            del v20
        if "c5table" in legs:
            # the same frames into a 1.68 GB table: 64 of them, short windows (a frame takes ~ 0.25 ms)
            n5 = min(len(c3_poses), WORKLOADS["C5table"]["frames"])
            c5_rec, c5_it, _, _ = measure_workload(args, V, L, synth, torch, "C5table", local_rank, 25, 5,
                                                   frames=(c3_poses[:n5], c3_verts[:n5]))
            c5_rec["workload"] = WORKLOADS["C5table"]["desc"]
            c5_rec["unit"] = "frames/s"
            if "index" in legs:
                c5_rec["occupancy_index_variant"] = index_variant_record(
                    args, c5_it, "C5table", 50, 10, lambda: (c5_it.sync(), torch.cuda.synchronize()))
            extra["configs"]["C5table"] = c5_rec
            c5_it.close()
        del c3_verts
        torch.cuda.empty_cache()
        fposes, fverts = render_frames(synth, wl, min(nframes, 64), dev, torch, sensor=False)     # for the CPU sample below

    # ---- the sharded path with ONE rank over RCCL: the code path the N > 1 lines run, so that the
    # scaling curve's anchor and its points share a code path ----
    if "sharded" in legs and name == "C2":
        try:
            if it is not None:
                it.close()
                it = None
            from voxelhashing_demo_amd import dist as vdist
            init_dist(dist, torch, local_rank)
            sargs = argparse.Namespace(**vars(args))
            # (windows of ~20 ms: with the 4 ms windows of earlier rounds the flush at each window's end cost the leg 3-4 %)
            sargs.steps, sargs.warmup = max(20, min(args.steps, 1000) // max(1, args.batch)), 5
            sargs.metric_name = baseline_metric(Wd, Ht)
            extra["sharded_world1"] = vdist.bench_sharded(sargs, wl, name, 0, 1, local_rank)
            if "index" in legs:
                # the same leg on the walk-free multi-camera frame (flatten_variant 4 on the shard: IndexSinkMulti)
                iargs = argparse.Namespace(**vars(sargs))
                iargs.option = list(getattr(sargs, "option", []) or []) + ["flatten_variant=4"]
                irec = vdist.bench_sharded(iargs, wl, name, 0, 1, local_rank)
                extra["sharded_world1"]["occupancy_index_variant"] = dict(
                    value=irec["value"], unit="frames/s", ms_per_step=irec["ms_per_step"], roofline=irec["roofline"],
                    note="vh_set_option(flatten_variant=4) on the shard: the multi-camera frame's walk over the bucket-occupancy "
                         "bitmap; not the reference's walk, hence not the leg's value")
            dist.destroy_process_group()
        except Exception as e:       # the main line must not be lost to a transport problem
            extra["sharded_world1"] = dict(error=repr(e))

    # ---- CPU baseline: the oracle on this box's host cores, bounded sample ----
    cpu = cpu_baseline_leg(args, name, fposes, fverts) if "cpu" in legs else None
    if "cpu" in legs:
        extra.setdefault("configs", {})["C1"] = dict(cpu_baseline=cpu_baseline_c1(synth), workload="C1: BASELINE.json configs[0] "
                                                     "(CPU-only plumbing case: no GPU leg; the GPU parity of the same scenes is "
                                                     "tests/test_gpu_parity.py and __graft_entry__.smoke)")

    out = dict(
        metric=baseline_metric(Wd, Ht),
        value=main_rec["value"], unit="frames/s", n_gpus=1, steps=args.steps, warmup=args.warmup,
        ms_per_step=main_rec["ms_per_step"], higher_is_better=True, scaling="weak",
        vs_baseline=None, dtype="f32", data="synthetic",
        metric_note="value = frames/s TSDF-integrated: a step is one vh_integrate_batch of frames_per_step consecutive "
                    "resident frames (pipelined: one launch per frame), windows of exactly K steps with a synchronisation "
                    "(which flushes the last frame) on both sides, median window; two_launch_frame = the same unpipelined; "
                    "the raycast half of the metric is raycast_mpix_per_s",
        frames_per_step=main_rec["frames_per_step"], ms_per_frame=main_rec["ms_per_frame"],
        windows=main_rec["windows"], timed_s=main_rec["timed_s"],
        window_min_ms=main_rec["window_min_ms"], window_max_ms=main_rec["window_max_ms"],
        config=dict(workload=wl["desc"], resident_frames=main_rec["resident_frames"], semantics="pinhole", pipelined=True,
                    occupied_blocks=occ, occupied_blocks_last_frame=main_rec["occupied_blocks_last_frame"],
                    allocated_blocks=main_rec["allocated_blocks"]),
        roofline=main_rec.get("roofline"), cpu_baseline=cpu,
        raycast_mpix_per_s=round(raycast_mpix, 1) if raycast_mpix else None,
        kernels=main_rec.get("kernels"),
        frame_algorithmic_bytes=main_rec["frame_algorithmic_bytes"],
        frame_algorithmic_gbs=main_rec["frame_algorithmic_gbs"],
        frame_frac_of_hbm_peak=main_rec["frame_frac_of_hbm_peak"],
    )
    out.update(extra)
    if it is not None:
        it.close()
    torch.cuda.synchronize()
    emit(out, compact_line(out))


COMPACT_LIMIT = 4096         # the driver keeps the last 8 KB of stdout: the line it parses must fit with room to spare


def _pick(d, *keys):
    """The named keys of a record that are present and not None (numbers and short strings only reach the line)."""
    return {k: d[k] for k in keys if isinstance(d, dict) and d.get(k) is not None}


def _roofline_numbers(rf):
    return _pick(rf, "bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "bytes_per_launch", "us_per_launch",
                 "rocprofv3_us_per_launch", "frac_at_rocprofv3_mean") if rf else None


def _cpu_numbers(cb):
    if not cb:
        return None
    out = _pick(cb, "value", "unit", "cores", "kind", "one_thread_frames_per_s")
    out["sample"] = cb.get("sample_short") or str(cb.get("sample", ""))[:120]
    return out


def _walk_free_numbers(rec):
    """The walk-free frame in the line: frames/s, the launch, and the latency/issue model it is held against (not an HBM fraction)."""
    if not rec:
        return None
    out = _pick(rec, "value", "ms_per_step")
    rf = rec.get("roofline") or {}
    out.update(_pick(rf, "us_per_launch", "traffic", "bytes_per_launch"))
    out["bound"] = "latency+issue"
    if rec.get("latency_model"):
        out["model_us"] = rec["latency_model"].get("model_us")
    return out


def compact_line(out):
    """The ONE stdout line of a one-GPU run: the contract's keys, the roofline and CPU baseline as numbers, and per comparison
    leg {value, ms_per_step, frac}.  Everything else (notes, formulas, per-kernel tables) is in bench_detail.json."""
    line = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                     "scaling", "vs_baseline", "dtype", "data")}
    line["config"] = _pick(out["config"], "workload", "resident_frames", "occupied_blocks", "allocated_blocks", "pipelined")
    line["config"]["frames_per_step"] = out.get("frames_per_step")
    line["roofline"] = _roofline_numbers(out.get("roofline"))
    if line["roofline"] is not None:
        line["roofline"].setdefault("traffic", None)
    line["cpu_baseline"] = _cpu_numbers(out.get("cpu_baseline"))
    line["raycast_mpix_per_s"] = out.get("raycast_mpix_per_s")
    line["windows"], line["timed_s"] = out.get("windows"), out.get("timed_s")
    rc = out.get("raycast")
    if rc:
        line["raycast"] = dict(_pick(rc, "kernel_us", "kernel_mpix_per_s", "poses"),
                               **({"valu_issue_frac": rc["roofline"].get("frac")} if rc.get("roofline") else {}))
    legs = {}
    for key, short in (("two_launch_frame", "two_launch"), ("first_lap", "first_lap")):
        if out.get(key):
            legs[short] = _pick(out[key], "value", "ms_per_step")
    if out.get("occupancy_index_variant"):
        legs["walk_free"] = _walk_free_numbers(out["occupancy_index_variant"])
    if out.get("loaded_integrate"):
        lr = out["loaded_integrate"]
        legs["loaded"] = dict(_pick(lr, "value", "ms_per_step", "occupied_blocks"),
                              **_pick(lr.get("roofline") or {}, "frac", "us_per_launch", "traffic", "frac_at_rocprofv3_mean"))
    for cname, crec in (out.get("configs") or {}).items():
        c = dict(_pick(crec, "value", "ms_per_step", "occupied_blocks"),
                 **_pick(crec.get("roofline") or {}, "frac", "us_per_launch", "traffic", "frac_at_rocprofv3_mean"))
        if crec.get("cpu_baseline"):
            c["cpu_baseline"] = _cpu_numbers(crec["cpu_baseline"])
        if crec.get("occupancy_index_variant"):
            c["walk_free"] = _walk_free_numbers(crec["occupancy_index_variant"])
        if crec.get("raycast"):
            c["raycast_us"] = dict(dda=crec["raycast"].get("kernel_us"), **(crec["raycast"].get("variants_kernel_us") or {}))
        legs[cname] = c
    sw = out.get("sharded_world1")
    if sw:
        legs["sharded_world1"] = (dict(error=str(sw["error"])[:160]) if "error" in sw else
                                  dict(_pick(sw, "value", "ms_per_step"), **_pick(sw.get("roofline") or {}, "frac", "us_per_launch")))
    if out.get("closed_loop"):
        legs["closed_loop"] = _pick(out["closed_loop"], "value", "unit", "integrate_us", "raycast_us", "align_us", "frames")
    line["legs"] = legs
    line["detail"] = "bench_detail.json"
    return line


def compact_sharded_line(out):
    """The ONE stdout line of an N-rank run (same rule: numbers and short names; the prose stays in bench_detail.json)."""
    line = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                     "scaling", "vs_baseline", "dtype", "data", "windows", "timed_s", "host_enqueue_ms_per_step")}
    cfg = out["config"]
    line["config"] = dict(workload=str(cfg["workload"])[:200],
                          **_pick(cfg, "frames_per_step", "frames_per_camera_per_exchange", "resident_frames", "pipelined",
                                  "key_bin_overflows", "occupied_blocks_all_ranks", "allocated_blocks_all_ranks", "packet_bytes",
                                  "key_bin_bytes_per_rank_and_exchange"))
    rf = out.get("roofline") or {}
    line["roofline"] = dict(_pick(rf, "bound", "kernel", "achieved", "peak", "unit", "frac", "bytes_per_launch", "us_per_launch",
                                  "launches_per_frame"), traffic=rf.get("traffic"))
    line["cpu_baseline"] = _cpu_numbers(out.get("cpu_baseline"))
    er = out.get("exchange_ranks") or {}
    line["exchange_ranks"] = dict(er, transport=str(er.get("transport", ""))[:120])
    ph = out.get("exchange_phases_us")
    line["exchange_phases_us"] = _pick(ph, "generate", "collectives", "apply", "first_to_last", "host_enqueue", "exchanges") if ph else None
    pr = out.get("predicted")
    if pr:
        line["predicted"] = {k: (_pick(v.get("nominal", v), "frames_per_s") if isinstance(v, dict) else v)
                             for k, v in pr.items() if k in ("reference_walk", "walk_free", "model")}
    line["exchange_host"] = str(out.get("exchange_host", ""))[:100]
    if out.get("generation_form"):
        line["generation_form"] = out["generation_form"]
    sr = out.get("sharded_raycast")
    if sr:
        line["sharded_raycast"] = _pick(sr, "mpix_per_s", "views_per_round", "ms_per_round", "lost_records")
    line["detail"] = "bench_detail.json"
    return line


def emit(detail, line):
    """Full record -> bench_detail.json (next to this file, and gpurun_out/ where that exists) and stderr; the compact line ->
    stdout, LAST: C stdio is flushed first (RCCL prints its version banner through it and would otherwise flush it at exit,
    behind the line), and nothing is printed after."""
    text = json.dumps(detail)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, "bench_detail.json"), "w") as f:
                    f.write(json.dumps(detail, indent=1) + "\n")
            except OSError:
                pass
    print("bench.py detail: " + text, file=sys.stderr, flush=True)
    s = json.dumps(line, separators=(",", ":"))
    if len(s) >= COMPACT_LIMIT:                 # never silently: a line the driver cannot parse is a lost round
        for k in ("legs", "predicted", "exchange_phases_us", "raycast"):
            line.pop(k, None)
            s = json.dumps(line, separators=(",", ":"))
            if len(s) < COMPACT_LIMIT:
                break
        print(f"bench.py: compact line trimmed to {len(s)} bytes", file=sys.stderr, flush=True)
    try:
        C.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stderr.flush()
    sys.stdout.write(s + "\n")
    sys.stdout.flush()


def init_dist(dist, torch, local_rank):
    if "RANK" not in os.environ or "WORLD_SIZE" not in os.environ:
        # direct (not torchrun) launch: one rank, any free port -- whatever MASTER_* a parent process may have left in the
        # environment (a test session that ran a process group of its own did)
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    # One node by contract.  NCCL_SOCKET_IFNAME names the interface of RCCL's BOOTSTRAP (and of its socket transport, which
    # GPUs of one node do not use: they talk over xGMI P2P / shared memory, chosen from the topology, not from this
    # variable).  The bootstrap of a one-node job is safest on the loop-back interface -- these containers have no network,
    # and their hostname may not resolve -- so it is the default here; a caller's own setting wins (setdefault), and the
    # shared-GPU rig below needs it (its ranks are "hosts" that can only meet over sockets on lo).
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    import datetime
    # VH_BENCH_BACKEND=gloo (with VH_BENCH_SHARE_GPU=1: every rank on cuda:0) is the test rig for the N > 1
    # code path on a one-GPU box -- RCCL refuses two ranks on one device; device buffers are then staged
    # through the host by the transport.  The driver's runs use the default: nccl = RCCL.
    backend = os.environ.get("VH_BENCH_BACKEND", "nccl")
    if backend == "nccl" and os.environ.get("VH_BENCH_SHARE_GPU") == "1":
        # the same rig on RCCL itself: RCCL only refuses two ranks of one HOST on one device, so every rank calls itself a
        # host of its own and the ranks meet over RCCL's socket transport on the loop-back interface -- a functional run of
        # the N > 1 path (tests/test_gpu_dist_rccl.py), never a measurement
        os.environ.setdefault("NCCL_HOSTID", f"voxelhash-bench-host-{os.environ['RANK']}")
        os.environ.setdefault("NCCL_IB_DISABLE", "1")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank),
                                timeout=datetime.timedelta(seconds=300))
    else:
        dist.init_process_group(backend, timeout=datetime.timedelta(seconds=300))


def raycast_roofline(workload, kernel_us, Wd, Ht):
    """The raycast is bound by VALU issue, not by bytes: instructions per wave come from
    the committed rocprofv3 --pmc pass (profiles/pmc_latest.json: SQ_INSTS_VALU / SQ_WAVES of
    raycast_kernel); the chip issues one VALU wave-instruction per SIMD per 2 cycles at most
    (MI355X_MICROARCH.md: a wave64 VALU op takes 2 cycles on a SIMD-32): 256 CUs x 4 SIMDs x 2.4 GHz / 2."""
    peak = 256 * 4 * 2.4e9 / 2 / 1e9                      # G wave-instructions / s
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json"))).get(workload + "_raycast", {})
        valu_per_wave = pmc.get("raycast_kernel_valu_per_wave")
        kernel = pmc.get("kernel", "raycast_kernel")
    except Exception:
        valu_per_wave, kernel = None, "raycast_dda_kernel"
    waves = math.ceil(Wd / 16) * math.ceil(Ht / 16) * 4
    if not valu_per_wave or kernel_us <= 0:
        return dict(bound="valu-issue", kernel=kernel, achieved=None, peak=round(peak, 1),
                    unit="G wave-instr/s", frac=None, traffic=None, note="no VALU count in profiles/pmc_latest.json")
    achieved = valu_per_wave * waves / (kernel_us * 1e-6) / 1e9
    return dict(bound="valu-issue", kernel=kernel, achieved=round(achieved, 1), peak=round(peak, 1),
                unit="G wave-instr/s", frac=round(achieved / peak, 4), traffic=None, valu_per_wave=valu_per_wave, waves=waves,
                us_per_launch=round(kernel_us, 2))


def closed_loop_leg(V, synth, torch, wl, poses, verts, local_rank, stream, nloop=60, first=200):
    """The KinectFusion loop of SURVEY.md 8(f)4 on this workload's first `nloop` frames, a FRESH table: per frame
    vh_preprocess -> vh_icp_align against the model's raycast maps -> vh_integrate_depth at the tracked pose ->
    vh_raycast_maps (tracking.FusionLoop; frame order of Application.cpp:73-90).  frames/s = tracked frames / wall time of
    the loop (one stream; the only host synchronisation of a frame is vh_icp_align handing back the pose).  The per-stage
    microseconds come from a second pass over the same frames with a stream synchronisation after each stage."""
    from voxelhashing_demo_amd import tracking
    Wd, Ht = wl["width"], wl["height"]
    # (the loop's first poses face one wall: a single plane leaves point-to-plane ICP singular -- tests/test_gpu_icp.py
    # test_single_plane_is_singular_on_the_gpu_too; from pose 200 on the view holds the room's corner and its objects)
    first = first if len(poses) >= first + nloop else 0
    poses, verts = poses[first:], verts[first:]
    n = min(nloop, len(poses))
    K = synth.K_matrix(Wd, Ht)
    k_inv = np.linalg.inv(K.astype(np.float64)).astype(np.float32)
    depth16 = torch.empty((n, Ht, Wd), dtype=torch.uint16, device=verts.device)
    for i in range(n):
        depth16[i] = (verts[i, :, :, 2] * 5000.0).round().clamp(0, 65535).to(torch.uint16)
    torch.cuda.synchronize()
    params = V.default_params(numBuckets=wl["buckets"], numVoxelBlocks=wl["blocks"], voxelSize=wl["voxel"])
    out = {}
    for timed_stages in (False, True):
        table = V.SDFHashtable(params, Wd, Ht, V.SEM_PINHOLE, device=local_rank, stream=stream)
        loop = tracking.FusionLoop(table, K, k_inv, stream=stream)
        with torch.cuda.stream(stream):
            loop.start(depth16[0], poses[0])
            table.synchronize()
            stage = dict(align=0.0, integrate=0.0, raycast=0.0)
            errs, rounds = [], 0
            t0 = time.perf_counter()
            for k in range(1, n):
                if not timed_stages:
                    loop.step(depth16[k])
                else:
                    c0 = time.perf_counter()
                    loop.track(depth16[k])                       # (synchronises by itself)
                    c1 = time.perf_counter()
                    p32 = loop.pose.astype(np.float32)
                    table.integrate_depth(p32, depth16[k], loop.k_inv)
                    table.synchronize()
                    c2 = time.perf_counter()
                    table.raycast_maps(p32, loop.depth, loop.model_v, loop.model_n)
                    table.synchronize()
                    c3 = time.perf_counter()
                    stage["align"] += c1 - c0
                    stage["integrate"] += c2 - c1
                    stage["raycast"] += c3 - c2
                rounds += loop.trk.iterations
                truth = np.asarray(poses[k], np.float64).reshape(4, 4)
                errs.append(float(np.abs(loop.pose[:3, 3] - truth[:3, 3]).max()))
            table.synchronize()
            wall = time.perf_counter() - t0
        if not timed_stages:
            out.update(value=round((n - 1) / wall, 1), unit="frames/s", frames=n - 1, ms_per_frame=round(1e3 * wall / (n - 1), 4),
                       icp_rounds_per_frame=round(rounds / (n - 1), 1), max_drift_mm=round(1e3 * max(errs), 2),
                       blocks=table.counters()["allocated_total"], first_pose=first)
        else:
            out.update(align_us=round(1e6 * stage["align"] / (n - 1), 1), integrate_us=round(1e6 * stage["integrate"] / (n - 1), 1),
                       raycast_us=round(1e6 * stage["raycast"] / (n - 1), 1))
        loop.close()
        table.close()
    out["note"] = ("tracking.FusionLoop on a fresh table: vh_preprocess -> vh_icp_align (up to 20 rounds, device-side solve, one "
                   "copy + synchronisation) -> vh_integrate_depth (two-launch frame: the model is read right after) -> vh_raycast_maps; "
                   "align_us includes vh_preprocess; the stage times are host-timed with a synchronisation per stage (second pass), "
                   "value is the unsynchronised loop; drift = max |translation error| against the synthetic truth")
    return out


def next_rows_leg(V, synth, torch, it, verts, k_inv, stream, Wd, Ht):
    from voxelhashing_demo_amd import tracking
    table, lib, h, pose_ptrs, nframes = it.table, it.lib, it.h, it.pose_ptrs, it.nframes
    dev = verts.device
    depth = torch.empty((Ht, Wd), dtype=torch.float32, device=dev)
    dptr = depth.data_ptr()
    # depth pre-processing (next #1): uint16 depth -> vertex + normal maps, one fused kernel
    depth_u16 = (verts[0, :, :, 2] * 5000.0).clamp(0, 65535).to(torch.uint16)
    pos_out, nrm_out = torch.empty_like(verts[0]), torch.empty_like(verts[0])
    torch.cuda.synchronize()
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
    # block silhouettes (row R1): front / back cube depth per pixel
    sil_f = torch.empty((Ht, Wd), dtype=torch.float32, device=dev)
    sil_b = torch.empty((Ht, Wd), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    for i in range(3):
        lib.vh_render_blocks(h, pose_ptrs[0], 0.1, 5.0, sil_f.data_ptr(), sil_b.data_ptr())
    table.synchronize()
    t6 = time.perf_counter()
    for i in range(20):
        lib.vh_render_blocks(h, pose_ptrs[(7 * i) % nframes], 0.1, 5.0, sil_f.data_ptr(), sil_b.data_ptr())
    table.synchronize()
    sil_us = 1e6 * (time.perf_counter() - t6) / 20
    table.set_profiling(True)
    for i in range(20):
        lib.vh_render_blocks(h, pose_ptrs[(7 * i) % nframes], 0.1, 5.0, sil_f.data_ptr(), sil_b.data_ptr())
    kt_s = table.kernel_times(reset=True)
    # garbage collection (next #4) over the blocks the last frame saw; threshold 0 frees them all
    it.frame(0)
    table.garbage_collect(0.0)
    kt_g = table.kernel_times(reset=True)
    gc_counters = table.counters()
    table.set_profiling(False)
    out = dict(
        block_silhouettes=dict(us_per_call=round(sil_us, 1), kernels_us=round(1e3 * kt_s["render_blocks_ms"] / 20, 1),
                               covered_pixels=int((sil_f > 0).sum()),
                               note="vh_render_blocks (SDFRenderer::drawToFrontAndBack): exact ray/box test of every "
                                    "allocated block's cube, host-timed per call and summed kernel time"),
        icp_round=dict(us_per_round=round(icp_us, 2), pairs=icp_sys[3], algorithmic_bytes=icp_bytes,
                       note="vh_icp_build_system against a raycast target, host-timed and synchronous "
                            "(the step API returns each round's 27 sums to the host)"),
        icp_align=dict(us_per_align=round(align_us, 1), rounds=trk.iterations,
                       us_per_round=round(align_us / max(1, trk.iterations), 2),
                       note="vh_icp_align: all rounds queued at once, 6x6 solve + SE3 update on the device, "
                            "one copy and one synchronisation at the end"),
        preprocess=dict(us_per_frame=round(pre_us, 2), algorithmic_bytes=pre_bytes,
                        achieved_gbs=round(pre_bytes / (pre_us * 1e-6) / 1e9, 1),
                        note="vh_preprocess, back-to-back calls timed on the host (launch gaps included)"),
        garbage_collect=dict(us_per_call=round(1e3 * kt_g["gc_ms"] / max(1, kt_g["gc_calls"]), 2),
                             blocks_freed=gc_counters["last_freed"],
                             note="vh_garbage_collect(0): identify + sweep + release + finish, "
                                  "every block of the last frame freed (8 KiB of voxel traffic each)"))
    trk.close()
    return out


if __name__ == "__main__":
    main()
