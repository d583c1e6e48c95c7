"""Bucket-range sharding of one logical voxel-hash table across the GPUs of a node
(SURVEY.md section 8(e)).

One process per GPU; rank r owns the buckets [r*per, (r+1)*per) of the logical
table -- the entries, the heap and the 4 KiB voxel blocks of every block key that
hashes there -- and holds camera r.  A step is a *multi-camera frame*: one frame
from every camera enters the table together.

    1. every rank: vertex map -> block keys (wave-deduplicated, frustum-tested),
       binned by owning rank, plus the camera packet (pose, inverse, camera-z plane)
    2. one all-to-all of the fixed-capacity key bins (count in the bin header, so no
       host synchronisation), one all-gather of the camera packets     [RCCL / xGMI]
    3. every rank, on its shard: new lock epoch, insert the received keys (camera
       order, then launch order, decides who wins a bucket), one walk over the shard
       for all cameras, TSDF update of every visible block in camera order

The per-rank logic (`sharded_step`) only talks to a *backend* (the HIP shard; tests plug in a
CPU oracle shard with the same interface, tests/oracle_shards.py) and a *transport* (torch.distributed, or an in-process
loop-back that plays all ranks in one process).  torch is transport and buffer owner only.
"""
from __future__ import annotations

import math
import os
import time

import numpy as np


# ----------------------------------------------------------------------------
# plan
# ----------------------------------------------------------------------------
class ShardPlan:
    """Bucket ranges of a logical table of `num_buckets` buckets cut over `world` ranks."""

    def __init__(self, num_buckets: int, world: int):
        if world < 1 or num_buckets < world:
            raise ValueError("need at least one bucket per rank")
        self.num_buckets, self.world = num_buckets, world
        self.per_shard = (num_buckets + world - 1) // world      # owner(h) = h // per_shard

    def bucket_range(self, rank: int):
        lo = rank * self.per_shard
        hi = min(self.num_buckets, lo + self.per_shard)
        if lo >= hi:
            raise ValueError(f"rank {rank} owns no bucket ({self.num_buckets} buckets over {self.world} ranks)")
        return lo, hi

    def owner(self, h: int) -> int:
        return h // self.per_shard


# ----------------------------------------------------------------------------
# transports
# ----------------------------------------------------------------------------
class TorchDistTransport:
    """torch.distributed collectives: backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)

    def _staged(self, t):
        """gloo moves host memory only: device buffers are staged through the host (debugging transport
        for several processes on ONE GPU, which RCCL refuses; never used with the nccl backend)."""
        return t.is_cuda and self.dist.get_backend(self.group) != "nccl"

    def all_to_all_bins(self, send, recv):
        """send/recv: [world, capacity, 4] int32; bin s of `send` goes to rank s."""
        if self._staged(send):
            h_send, h_recv = send.cpu().view(-1), recv.cpu().view(-1)
            self.dist.all_to_all_single(h_recv, h_send, group=self.group)
            recv.copy_(h_recv.view(recv.shape))
            return recv
        self.dist.all_to_all_single(recv.view(-1), send.view(-1), group=self.group)
        return recv

    def all_gather_packets(self, packet, out):
        """packet: [P] float32 -> out: [world, P], camera order = rank order."""
        if self._staged(packet):
            h_out = out.cpu()
            self.dist.all_gather(list(h_out.unbind(0)), packet.cpu(), group=self.group)
            out.copy_(h_out)
        elif packet.is_cuda:
            self.dist.all_gather_into_tensor(out.view(-1), packet, group=self.group)
        else:
            self.dist.all_gather(list(out.unbind(0)), packet, group=self.group)
        return out


    # ---- raycast over shards ----
    def _device(self):
        import torch
        return torch.device("cuda", torch.cuda.current_device()) if self.dist.get_backend(self.group) == "nccl" \
            else torch.device("cpu")

    def all_gather_poses(self, pose):
        """This rank's view pose -> [world, 16] float32 (host), view order = rank order."""
        import torch
        mine = torch.from_numpy(np.ascontiguousarray(np.asarray(pose, np.float32).reshape(16))).to(self._device())
        out = torch.empty((self.world, 16), dtype=torch.float32, device=mine.device)
        self.all_gather_packets(mine, out)
        return out.cpu().numpy()

    def exchange_view_records(self, send, counts, capacity, recv):
        """send: packed records [*, 4112] uint8, view (= rank) 0's first; counts: [world] int32 selected
        per view, on the same device (more than `capacity` means the excess was not written).  The
        clipped counts are exchanged first (one small all-to-all; the payload all-to-all needs them on
        the host: the only synchronisation of the round).  Records land in `recv` in source order.
        Returns (recv_counts, lost)."""
        import torch
        staged = self._staged(send)
        if staged:
            counts = counts.cpu()
        sc = torch.clamp(counts, max=capacity).to(torch.int64)
        rc = torch.empty_like(sc)
        self.dist.all_to_all_single(rc, sc, group=self.group)
        host = torch.stack([counts.to(torch.int64), sc, rc]).cpu()
        send_counts, recv_counts = host[1].tolist(), host[2].tolist()
        lost = int((host[0] - host[1]).sum())
        n_in, n_out = sum(send_counts), sum(recv_counts)
        if n_out > recv.shape[0]:
            raise RuntimeError(f"view receive buffer holds {recv.shape[0]} records, {n_out} arrive")
        if staged:
            h_recv = torch.empty((n_out, recv.shape[1]), dtype=recv.dtype)
            self.dist.all_to_all_single(h_recv, send[:n_in].cpu(), output_split_sizes=recv_counts,
                                        input_split_sizes=send_counts, group=self.group)
            recv[:n_out].copy_(h_recv)
            return recv_counts, lost
        self.dist.all_to_all_single(recv[:n_out], send[:n_in], output_split_sizes=recv_counts,
                                    input_split_sizes=send_counts, group=self.group)
        return recv_counts, lost


class LoopbackExchange:
    """All ranks of a sharded run inside ONE process (tests, single-GPU emulation): every
    virtual rank posts its send buffers, then each fetches what the collectives would deliver."""

    def __init__(self, world: int):
        self.world = world
        self.bins = [None] * world
        self.packets = [None] * world

    def post(self, rank: int, bins, packet):
        self.bins[rank], self.packets[rank] = bins.clone(), packet.clone()

    def fetch(self, rank: int, bins_recv, packets_out):
        import torch
        bins_recv.copy_(torch.stack([self.bins[src][rank] for src in range(self.world)]))     # all-to-all
        packets_out.copy_(torch.stack(self.packets))                                           # all-gather


# ----------------------------------------------------------------------------
# backends
# ----------------------------------------------------------------------------
# Buffers carry `batch` frames per camera so that one all-to-all and one all-gather move
# several multi-camera frames (fewer, larger collectives: their latency and the host cost of
# issuing them are paid once per batch).  The frames of a batch are still applied one
# multi-camera frame after the other, in order, so results do not depend on the batch size.
#   bins_send[dst, b], bins_recv[src, b] : [capacity, 4] int32 key bins
#   packet[b], packets[cam, b]           : [32 + W*H] float32 camera packets
class HipShard:
    """This rank's shard on its GPU (libvoxelhash_hip.so through the C-ABI)."""

    def __init__(self, params, width, height, semantics, plan: ShardPlan, rank: int, capacity: int,
                 batch: int = 1, device=None, stream=None, sets: int = 1, batched_calls: bool = True,
                 sensor_k_inv=None, per_batch_bins: bool = False):
        """sensor_k_inv: K^-1 (3x3) of the camera -> packets carry the uint16 sensor image (VH_PACKET_U16,
        half the bytes of the float camera-z plane); generate_* then also need the depth images.
        per_batch_bins: one key bin of `capacity` records per (owner, batch) instead of per (owner, frame)
        (VH_BIN_PER_BATCH; batched calls only): bins_send / bins_recv are [R, 1, capacity, 4]."""
        import torch

        from .hashtable import SDFHashtable
        self.plan, self.rank, self.capacity, self.batch = plan, rank, capacity, batch
        self.sensor_k_inv = None if sensor_k_inv is None else np.ascontiguousarray(
            np.asarray(sensor_k_inv, np.float32).reshape(9))
        self.batched_calls = batched_calls     # False: per-frame step calls (4 + 2 launches per frame)
        self.per_batch_bins = bool(per_batch_bins)
        assert batched_calls or not per_batch_bins
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        self.table = SDFHashtable(params, width, height, semantics, device=self.device.index,
                                  bucket_range=plan.bucket_range(rank), stream=stream)
        if self.sensor_k_inv is not None:
            assert (width * height) % 2 == 0
            self.table.set_option("packet_format", 1)
            self.packet_floats = P = 36 + width * height // 2
        else:
            self.packet_floats = P = 32 + width * height
        R, B = plan.world, batch
        BB = 1 if self.per_batch_bins else B           # bins per (owner, exchange)
        # `sets` independent buffer sets: the pipelined step fills one while the other is consumed
        self.sets = []
        for _ in range(max(1, sets)):
            s = dict(bins_send=torch.zeros((R, BB, capacity, 4), dtype=torch.int32, device=self.device),
                     bins_recv=torch.zeros((R, BB, capacity, 4), dtype=torch.int32, device=self.device),
                     packet=torch.zeros((B, P), dtype=torch.float32, device=self.device),
                     packets=torch.zeros((R, B, P), dtype=torch.float32, device=self.device))
            s["send_b"] = [s["bins_send"][0, b].data_ptr() for b in range(BB)]
            s["recv_b"] = [s["bins_recv"][0, b].data_ptr() for b in range(BB)]
            s["packet_b"] = [s["packet"][b].data_ptr() for b in range(B)]
            s["packets_b"] = [s["packets"][0, b].data_ptr() for b in range(B)]
            self.sets.append(s)
        self.use_set(0)
        # the buffers were zero-filled on torch's current stream and will be used from other streams
        # (torch streams do not order themselves against each other): make them ready for all
        torch.cuda.synchronize(self.device)

    def use_set(self, i: int):
        s = self.sets[i]
        self.bins_send, self.bins_recv, self.packet, self.packets = (
            s["bins_send"], s["bins_recv"], s["packet"], s["packets"])
        self._cur = s

    def generate(self, b: int, pose, verts, depth=None):
        import ctypes as C
        self.table.set_pose(pose)
        if self.sensor_k_inv is None:
            self.table.generate_keys(verts, self.rank, self.plan.world, self._cur["send_b"][b], self.capacity,
                                     self._cur["packet_b"][b], bin_stride=self.batch * self.capacity)
            return
        self.table.generate_keys(verts, self.rank, self.plan.world, self._cur["send_b"][b], self.capacity, None,
                                 bin_stride=self.batch * self.capacity)
        p16 = np.ascontiguousarray(np.asarray(pose, np.float32).reshape(1, 16))
        self.table.write_packets_u16_batch(p16, (C.c_void_p * 1)(depth.data_ptr()), self.sensor_k_inv,
                                           self._cur["packet_b"][b], 1)

    def apply(self, b: int):
        self.table.reset_mutexes()
        self.table.insert_bins(self._cur["recv_b"][b], self.plan.world, self.capacity,
                               bin_stride=self.batch * self.capacity)
        self.table.integrate_packets(self.plan.world, self._cur["packets_b"][b],
                                     packet_stride=self.batch * self.packet_floats)

    # whole batch in one C call each (fewest launches: 1 + batch, and 2 per multi-camera frame)
    def generate_all(self, poses, verts_list, depth_list=None):
        import ctypes as C
        if not self.batched_calls:
            for b in range(self.batch):
                self.generate(b, poses[b], verts_list[b], None if depth_list is None else depth_list[b])
            return
        p16 = np.ascontiguousarray(np.asarray(poses, np.float32).reshape(self.batch, 16))
        if self.sensor_k_inv is None:
            ptrs = (C.c_void_p * self.batch)(*[v.data_ptr() for v in verts_list])
            self.table.generate_keys_batch(p16, ptrs, self.rank, self.plan.world, self.bins_send, self.capacity,
                                           self.packet, self.batch, self.per_batch_bins)
            return
        # keys and packets from the sensor images alone, one launch per 8 frames (verts_list is not read)
        dptrs = (C.c_void_p * self.batch)(*[d.data_ptr() for d in depth_list])
        self.table.generate_keys_depth_batch(p16, dptrs, self.sensor_k_inv, self.rank, self.plan.world, self.bins_send,
                                             self.capacity, self.packet, self.batch, self.per_batch_bins)

    def apply_all(self):
        if not self.batched_calls:
            for b in range(self.batch):
                self.apply(b)
            return
        self.table.apply_frames_batch(self.bins_recv, self.plan.world, self.capacity, self.plan.world, self.packets,
                                      self.batch, self.per_batch_bins)


    # raycast over shards: this shard's blocks that each of the world's views can touch
    def export_views(self, poses, capacity: int, t_min: float = 0.1, t_max: float = 5.0):
        """-> (records [world*capacity, 4112] uint8, view 0's first, packed; counts [world] int32), on the device."""
        import torch
        n = len(poses)
        if getattr(self, "_view_send", None) is None or self._view_send.shape[0] < n * capacity:
            self._view_send = torch.zeros((n * capacity, VIEW_RECORD_BYTES), dtype=torch.uint8, device=self.device)
            self._view_counts = torch.zeros((n,), dtype=torch.int32, device=self.device)
        self.table.export_views(poses, self._view_send, capacity, self._view_counts, t_min, t_max)
        return self._view_send, self._view_counts


# ----------------------------------------------------------------------------
# the native host (include/voxelhash_dist.h): the same pipeline inside the library, on RCCL directly
# ----------------------------------------------------------------------------
def unique_id(rank: int = 0, broadcast=None) -> bytes:
    """The 128 bytes every rank hands to NativeDist: rank 0 draws them (ncclGetUniqueId), `broadcast(bytes or None)
    -> bytes` carries them to the others (torch.distributed, MPI, a file ...).  One rank: no exchange."""
    import ctypes as C

    from . import _lib as L
    buf = None
    if rank == 0:
        raw = C.create_string_buffer(128)
        L.check(L.load().vh_dist_unique_id(raw), "vh_dist_unique_id")
        buf = raw.raw
    return broadcast(buf) if broadcast is not None else buf


def loopback_id() -> bytes:
    """The id of a fresh loop-back group (include/voxelhash_dist.h: vh_dist_loopback_id): `world` NativeDist instances of
    THIS process created with it exchange through hipMemcpyAsync instead of RCCL; their collective calls must be made
    from one host thread per rank (NativeGroup does)."""
    import ctypes as C

    from . import _lib as L
    raw = C.create_string_buffer(128)
    L.check(L.load().vh_dist_loopback_id(raw), "vh_dist_loopback_id")
    return raw.raw


def torch_broadcast_bytes(group=None):
    """A `broadcast` for unique_id over torch.distributed (any backend)."""
    def bc(buf):
        import torch
        import torch.distributed as dist
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
        t = torch.zeros(128, dtype=torch.uint8) if buf is None else torch.frombuffer(bytearray(buf), dtype=torch.uint8)
        t = t.to(dev)
        dist.broadcast(t, 0, group=group)
        return bytes(t.cpu().numpy().tobytes())
    return bc


class NativeDist:
    """This rank of the sharded table, driven through ONE C call per exchange: generation, the RCCL all-to-all of the key
    bins and all-gather of the packets, and the application run inside libvoxelhash_hip.so on three HIP streams
    (include/voxelhash_dist.h).  What ShardedPipeline + HipShard + TorchDistTransport do from Python, minus a dozen
    interpreter round trips and four torch collectives per exchange."""

    def __init__(self, params, width, height, semantics, rank: int, world: int, batch: int, uid: bytes,
                 sensor_k_inv=None, key_capacity: int = 0, device: int = -1):
        import ctypes as C

        from . import _lib as L
        from .hashtable import SDFHashtable
        self._L, self._lib = L, L.load()
        cfg = L.DistConfig()
        cfg.table = L.Config(params, width, height, semantics, device)
        cfg.rank, cfg.world, cfg.batch, cfg.key_capacity = rank, world, batch, key_capacity
        cfg.packet_format = 1 if sensor_k_inv is not None else 0
        if sensor_k_inv is not None:
            cfg.k_inv = (C.c_float * 9)(*np.asarray(sensor_k_inv, np.float32).reshape(9))
        h = C.c_void_p()
        L.check(self._lib.vh_dist_create(C.byref(cfg), uid, None, C.byref(h)), "vh_dist_create")
        self._h, self.rank, self.world, self.batch = h, rank, world, batch
        plan = ShardPlan(params.numBuckets, world)
        self.table = SDFHashtable.borrowed(self._lib.vh_dist_shard(h), params, width, height, semantics, plan.bucket_range(rank))
        self.transport = self._lib.vh_dist_transport_name(h).decode()

    def order_against(self, stream=None):
        """Orders step() / raycast() against a torch stream (default: the current one) like ordinary stream work: frames are
        read behind what the stream has queued and may be overwritten by what it queues afterwards; the raycast image is
        ready for work queued on the stream after the call (include/voxelhash_dist.h, STREAM CONTRACT).  None of it costs a
        host synchronisation.  `stream=False` returns to the caller-synchronises contract."""
        import torch
        if stream is False:
            self._L.check(self._lib.vh_dist_set_user_stream(self._h, None, 0), "vh_dist_set_user_stream")
            return
        st = torch.cuda.current_stream() if stream is None else stream
        self._L.check(self._lib.vh_dist_set_user_stream(self._h, st.cuda_stream, 1), "vh_dist_set_user_stream")

    def step(self, poses, frames):
        """One exchange: `batch` poses and device tensors (uint16 sensor images, or float4 vertex maps) of THIS rank's camera."""
        import ctypes as C
        p16 = np.ascontiguousarray(np.asarray(poses, np.float32).reshape(self.batch, 16))
        ptrs = (C.c_void_p * self.batch)(*[f.data_ptr() for f in frames])
        self.step_raw(p16.ctypes.data_as(C.POINTER(C.c_float)), ptrs)

    def step_raw(self, pose_ptr, frame_ptrs):
        rc = self._lib.vh_dist_step_batch(self._h, pose_ptr, frame_ptrs)
        if rc != 0:
            self._L.check(rc, "vh_dist_step_batch")

    def flush(self):
        self._L.check(self._lib.vh_dist_flush(self._h), "vh_dist_flush")

    def raycast(self, pose, out, capacity: int, t_min: float = 0.1, t_max: float = 5.0, lost=None):
        import ctypes as C
        p = np.ascontiguousarray(np.asarray(pose, np.float32).reshape(16))
        self._L.check(self._lib.vh_dist_raycast(self._h, p.ctypes.data_as(C.POINTER(C.c_float)), t_min, t_max, capacity,
                                                out.data_ptr(), None if lost is None else lost.data_ptr()), "vh_dist_raycast")
        return out

    def raycast_auto(self, pose, out, normals=None, t_min: float = 0.1, t_max: float = 5.0):
        """The raycast round with the slot capacity agreed by the ranks (vh_dist_raycast_auto): repeated with more room while
        any rank's view lost records; returns the capacity that rendered every view whole.  Synchronises.  Collective."""
        import ctypes as C
        p = np.ascontiguousarray(np.asarray(pose, np.float32).reshape(16))
        cap = C.c_int32()
        self._L.check(self._lib.vh_dist_raycast_auto(self._h, p.ctypes.data_as(C.POINTER(C.c_float)), t_min, t_max, out.data_ptr(),
                                                     None if normals is None else normals.data_ptr(), C.byref(cap)), "vh_dist_raycast_auto")
        return cap.value

    def set_option(self, name: str, value: int):
        """vh_dist_set_option: "force_collectives" (a one-rank group runs the collectives anyway), "phase_timing"."""
        self._L.check(self._lib.vh_dist_set_option(self._h, name.encode(), int(value)), "vh_dist_set_option")

    def phase_times(self, reset: bool = True):
        """Per-exchange phase times in microseconds (option "phase_timing"): means over the exchanges completed since the last reset."""
        ph = self._L.DistPhases()
        self._L.check(self._lib.vh_dist_phase_times(self._h, ph, 1 if reset else 0), "vh_dist_phase_times")
        n = max(1, ph.exchanges)
        return dict(exchanges=int(ph.exchanges), generate=ph.generate_us / n, collectives=ph.collectives_us / n, apply=ph.apply_us / n,
                    first_to_last=ph.first_to_last_us / n, host_enqueue=ph.host_enqueue_us / n)

    def self_check(self):
        """vh_dist_self_check: a known pattern through the transport's all-to-all and all-gather, compared on the device.  Collective."""
        self._L.check(self._lib.vh_dist_self_check(self._h), "vh_dist_self_check")

    def comm_info(self):
        """(rank, size) as the transport reports them (ncclCommUserRank / ncclCommCount over RCCL)."""
        import ctypes as C
        r, n = C.c_int32(), C.c_int32()
        self._L.check(self._lib.vh_dist_comm_info(self._h, C.byref(r), C.byref(n)), "vh_dist_comm_info")
        return r.value, n.value

    def generation_form(self) -> str:
        """"fused" (the key generation rides in the frame launches) or "separate" (launches of its own): this rank's choice."""
        f = self._lib.vh_dist_generation_form(self._h)
        if f < 0:
            self._L.check(-f, "vh_dist_generation_form")
        return "fused" if f else "separate"

    def host_stats(self):
        import ctypes as C
        s, n = C.c_double(), C.c_uint64()
        self._L.check(self._lib.vh_dist_host_stats(self._h, C.byref(s), C.byref(n)), "vh_dist_host_stats")
        return s.value, n.value

    def close(self):
        if getattr(self, "_h", None):
            self.table.close()
            self._lib.vh_dist_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class NativeGroup:
    """All `world` ranks of the native exchange in ONE process on ONE GPU, joined by the library's loop-back transport
    (vh_dist_loopback_id) and driven by one host thread per rank -- the stand-in for `world` processes on `world` GPUs that
    lets vh_dist_step_batch / vh_dist_raycast (the code bench.py --gpus N runs) execute with R > 1 on a single-GPU box.
    ctypes releases the GIL inside the C calls, so the R calls of a collective really are concurrent."""

    def __init__(self, params, width, height, semantics, world: int, batch: int, sensor_k_inv=None, key_capacity: int = 0,
                 device: int = -1, options=None, band: float = 0.0):
        from concurrent.futures import ThreadPoolExecutor
        uid = loopback_id()
        self.world, self.batch = world, batch
        self.ranks = [NativeDist(params, width, height, semantics, r, world, batch, uid, sensor_k_inv=sensor_k_inv,
                                 key_capacity=key_capacity, device=device) for r in range(world)]
        for nd in self.ranks:
            assert nd.transport == "loopback"
            for k, v in (options or {}).items():
                nd.table.set_option(k, v)
            if band > 0.0:
                nd.table.set_alloc_band(band)
        self._pool = ThreadPoolExecutor(max_workers=world)

    def _all(self, fn):
        futures = [self._pool.submit(fn, r, nd) for r, nd in enumerate(self.ranks)]
        errors, out = [], []
        for f in futures:
            try:
                out.append(f.result())
            except Exception as e:          # noqa: BLE001 -- every rank's call is awaited before the first error is raised
                errors.append(e)
        if errors:
            raise errors[0]
        return out

    def step(self, poses, frames):
        """One exchange: poses[r][b], frames[r][b] for every rank r and frame b of the batch."""
        self._all(lambda r, nd: nd.step(poses[r], frames[r]))

    def raycast(self, poses, outs, capacity: int, t_min: float = 0.1, t_max: float = 5.0, losts=None):
        """One raycast round: rank r renders poses[r] into outs[r]."""
        self._all(lambda r, nd: nd.raycast(poses[r], outs[r], capacity, t_min, t_max, None if losts is None else losts[r]))
        return outs

    def flush(self):
        for nd in self.ranks:
            nd.flush()

    def self_check(self):
        """vh_dist_self_check on every rank (collective): the transport carries a known pattern to the right places."""
        self._all(lambda r, nd: nd.self_check())

    @property
    def tables(self):
        return [nd.table for nd in self.ranks]

    def close(self):
        if getattr(self, "_pool", None) is not None:
            self._pool.shutdown(wait=True)
            self._pool = None
        for nd in self.ranks:          # (every rank idle before any is destroyed: the loop-back events are shared, voxelhash_dist.h)
            try:
                nd.flush()
            except Exception:          # noqa: BLE001 -- a rank that already failed is destroyed all the same
                pass
        for nd in self.ranks:
            nd.close()


# ----------------------------------------------------------------------------
# raycast over shards: a ray samples blocks of every shard, so the rank that
# renders a view gathers the blocks the view can touch and raycasts a private view table.
#   1. all-gather of the view poses (every rank renders its own camera's view)
#   2. every rank: ONE walk over its shard for all views -> records {key, 512 voxels} per view
#   3. all-to-all of the per-view record counts, then of the records          [RCCL / xGMI]
#   4. every rank: import into its view table, vh_raycast on it
# The selection is a conservative superset of the blocks the rays sample, so the depth image is
# bit-equal to a raycast of the unsharded table.  (Compositing per-shard raycasts by min depth,
# the other option of 8(e), loses every surface crossing whose two samples lie in blocks of
# different owners: 4 % holes at 4 ranks in the room scene, measured with the oracle.)
# ----------------------------------------------------------------------------
VIEW_RECORD_BYTES = 4112


def _view_params(params):
    p = type(params).from_buffer_copy(params)
    p.numVoxelBlocks = 1            # the voxels of a view table stay in the received records
    return p


class HipViewTable:
    """Rendering side on the GPU: a dedicated unsharded context plus the receive buffer."""

    def __init__(self, params, width, height, semantics, world: int, capacity: int, device=None, stream=None):
        import torch

        from .hashtable import SDFHashtable
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        self.table = SDFHashtable(_view_params(params), width, height, semantics, device=self.device.index,
                                  stream=stream)
        self.recv = torch.zeros((world * capacity, VIEW_RECORD_BYTES), dtype=torch.uint8, device=self.device)
        self.depth = torch.zeros((height, width), dtype=torch.float32, device=self.device)
        torch.cuda.synchronize(self.device)      # ready for whichever stream uses them

    def render(self, count: int, pose, t_min: float = 0.1, t_max: float = 5.0):
        self.table.import_view(self.recv, count)
        return self.table.raycast(pose, self.depth, t_min, t_max)


def _clip_counts(counts, capacity):
    """Host copy of the per-view counts (synchronises); (sent, lost) records per view."""
    demanded = [int(x) for x in counts.cpu().tolist()]
    return [min(d, capacity) for d in demanded], [max(0, d - capacity) for d in demanded]


def sharded_raycast(shard, view, transport: TorchDistTransport, pose, capacity: int,
                    t_min: float = 0.1, t_max: float = 5.0):
    """Depth image of this rank's view `pose` through the whole sharded table.  Returns
    (depth, lost): `lost` > 0 means some shard selected more than `capacity` blocks for a view and
    the image may miss surfaces (raise the capacity)."""
    # Everything of the round is enqueued on torch's CURRENT stream: the buffers are torch tensors and the
    # collectives run there, so the two contexts must not keep enqueuing on streams of their own (torch
    # streams do not synchronise with each other implicitly: an export on another stream raced with the
    # zero-fill and the host read of its counts).  Pending work of the contexts is waited for first.
    restore = []
    if hasattr(shard.table, "stream_handle"):
        import torch
        cur = torch.cuda.current_stream().cuda_stream
        for t in (shard.table, view.table):
            if t.stream_handle != cur:
                t.synchronize()
                restore.append((t, t.stream_handle))
                t.set_stream(cur)
    poses = transport.all_gather_poses(pose)
    records, counts = shard.export_views(poses, capacity, t_min, t_max)
    recv_counts, lost = transport.exchange_view_records(records, counts, capacity, view.recv)   # synchronises
    depth = view.render(sum(recv_counts), pose, t_min, t_max)
    for t, handle in restore:
        t.synchronize()                  # its work of this round is done before it goes back to its own stream
        t.set_stream(handle)
    return depth, lost


def sharded_raycast_fixed(shard, view, transport: TorchDistTransport, pose, capacity: int,
                          t_min: float = 0.1, t_max: float = 5.0, state=None):
    """The same round with NO host synchronisation: the poses are gathered on the device, every shard
    exports view v's records into the fixed slot range [v*capacity, (v+1)*capacity), the exchange has
    equal sizes (records: world x capacity x 4112 bytes per rank, counts: world int32), and the view
    table reads the counts on the device.  Trades bandwidth for latency: the payload does not shrink
    with the number of blocks a view really touches, so `capacity` should be sized to the scene.
    Returns (depth, lost) with `lost` a DEVICE tensor (records the shards selected beyond the capacity,
    summed over this view's sources) that the caller may inspect whenever it synchronises anyway.
    `state`: dict kept by the caller across rounds (buffers)."""
    import torch
    world = transport.world
    dev = shard.device
    restore = []
    cur = torch.cuda.current_stream().cuda_stream
    for t in (shard.table, view.table):
        if t.stream_handle != cur:
            t.synchronize()
            restore.append((t, t.stream_handle))
            t.set_stream(cur)
    st = state if state is not None else {}
    if st.get("capacity") != capacity or st.get("world") != world:
        st.update(capacity=capacity, world=world,
                  pose_all=torch.empty((world, 16), dtype=torch.float32, device=dev),
                  pose_mine=torch.empty((16,), dtype=torch.float32, device=dev),
                  send=torch.zeros((world * capacity, VIEW_RECORD_BYTES), dtype=torch.uint8, device=dev),
                  recv=torch.zeros((world * capacity, VIEW_RECORD_BYTES), dtype=torch.uint8, device=dev),
                  counts=torch.zeros((world,), dtype=torch.int32, device=dev))
    # From PAGEABLE host memory: the runtime stages the 64 bytes before the call returns, so the next round may
    # overwrite its own pose at once.  (A single pinned staging buffer copied with non_blocking=True is read by the
    # DMA whenever the stream gets there -- back-to-back rounds then exported the blocks of the NEXT round's view.)
    st["pose_mine"].copy_(torch.from_numpy(np.ascontiguousarray(np.asarray(pose, np.float32).reshape(16))))
    transport.dist.all_gather_into_tensor(st["pose_all"].view(-1), st["pose_mine"], group=transport.group)
    shard.table.export_views_fixed(st["pose_all"], world, st["send"], capacity, st["counts"], t_min, t_max)
    # one payload collective: the counts ride in the spare header word of each slot range's first record
    transport.dist.all_to_all_single(st["recv"].view(-1), st["send"].view(-1), group=transport.group)
    view.table.import_views(st["recv"], world, capacity, None)
    depth = view.table.raycast(pose, view.depth, t_min, t_max)
    heads = st["recv"].view(world, capacity * VIEW_RECORD_BYTES)[:, 12:16].contiguous().view(torch.int32).view(-1)
    lost = torch.clamp(heads - capacity, min=0).sum()
    for t, handle in restore:
        t.synchronize()
        t.set_stream(handle)
    return depth, lost


def loopback_raycast_fixed(shards, views, poses, capacity: int, t_min: float = 0.1, t_max: float = 5.0):
    """vh_export_views_fixed / vh_import_views with every rank played in this process."""
    import torch
    world = len(shards)
    dev = shards[0].device
    d_poses = torch.from_numpy(np.ascontiguousarray(np.asarray(poses, np.float32).reshape(world, 16))).to(dev)
    sends, counts = [], []
    for sh in shards:
        send = torch.zeros((world * capacity, VIEW_RECORD_BYTES), dtype=torch.uint8, device=dev)
        cnt = torch.zeros((world,), dtype=torch.int32, device=dev)
        sh.table.export_views_fixed(d_poses, world, send, capacity, cnt, t_min, t_max)
        sends.append(send.view(world, capacity, VIEW_RECORD_BYTES))
        counts.append(cnt)
    out = []
    for r, view in enumerate(views):
        recv = torch.stack([sends[src][r] for src in range(world)]).view(world * capacity, VIEW_RECORD_BYTES).contiguous()
        cin = torch.stack([counts[src][r] for src in range(world)]).contiguous()
        torch.cuda.synchronize()
        view.table.import_views(recv, world, capacity, cin if r % 2 == 0 else None)   # counts array / header word
        view._fixed_recv = recv
        depth = view.table.raycast(poses[r], view.depth, t_min, t_max)
        torch.cuda.synchronize()
        assert int(torch.clamp(cin - capacity, min=0).sum()) == 0, "view capacity exceeded"
        out.append(depth.cpu().numpy().copy())
    return out


def loopback_raycast(shards, views, poses, capacity: int, t_min: float = 0.1, t_max: float = 5.0):
    """The same with every rank played in this process: view r is rendered from poses[r] by views[r].
    Returns the depth images as numpy arrays."""
    import torch
    world = len(shards)
    exports = []
    for sh in shards:
        records, counts = sh.export_views(poses, capacity, t_min, t_max)
        sent, lost = _clip_counts(counts, capacity)
        assert sum(lost) == 0, "view capacity exceeded"
        exports.append((records, sent))
    out = []
    for r, view in enumerate(views):
        parts = []
        for src in range(world):
            records, sent = exports[src]
            first = sum(sent[:r])
            parts.append(records[first:first + sent[r]])
        got = torch.cat(parts)
        if got.shape[0] > view.recv.shape[0]:
            view.recv = torch.zeros((got.shape[0], VIEW_RECORD_BYTES), dtype=torch.uint8, device=got.device)
        view.recv[:got.shape[0]].copy_(got)
        depth = view.render(got.shape[0], poses[r], t_min, t_max)
        if depth.is_cuda:
            torch.cuda.synchronize()
        out.append(depth.cpu().numpy().copy())
    return out


# ----------------------------------------------------------------------------
# the step
# ----------------------------------------------------------------------------
def sharded_step(shard, transport: TorchDistTransport, poses, verts_list, depth_list=None):
    """`batch` multi-camera frames from this rank's point of view: poses[b], verts_list[b] (and, for
    sensor-depth packets, depth_list[b]) are this rank's camera for frame b of the batch."""
    assert len(poses) == shard.batch == len(verts_list)
    shard.generate_all(poses, verts_list, depth_list)
    transport.all_to_all_bins(shard.bins_send, shard.bins_recv)
    transport.all_gather_packets(shard.packet.view(-1), shard.packets.view(shard.plan.world, -1))
    shard.apply_all()


class ShardedPipeline:
    """The same steps, software-pipelined over three HIP streams (plus RCCL's own): while the table
    stream applies exchange i (insert / walk / TSDF update), the collectives of exchange i+1 are in
    flight and the keys and packets of exchange i+2 are being generated, so an exchange costs
    max(apply, RCCL, generate) instead of their sum.  Two buffer sets alternate; events order the
    hand-offs:

        gen:   [wait ready(i-2): the send buffers of this set have been sent]  generate(i)  [record generated(i)]
        front: [wait generated(i), applied(i-2): the receive buffers are free]  all_to_all, all_gather  [record ready(i)]
        table: [wait ready(i)]  apply(i)  [record applied(i)]

    (generate only writes a set's send buffers and apply only reads its receive buffers, so exchange
    i+2 may be generated while exchange i is still being applied.)  Operations on the table are issued
    in exactly the order of `sharded_step`, so results are the same.  `shard` needs sets=2."""

    def __init__(self, shard: HipShard, transport: TorchDistTransport, table_stream, front_stream, gen_stream=None):
        import torch
        assert len(shard.sets) >= 2
        self.torch, self.shard, self.transport = torch, shard, transport
        self.table_stream, self.front_stream = table_stream, front_stream
        self.gen_stream = gen_stream if gen_stream is not None else torch.cuda.Stream(device=shard.device)
        self.generated = [torch.cuda.Event() for _ in range(2)]  # send buffers of set s are filled
        self.ready = [torch.cuda.Event() for _ in range(2)]      # exchange of set s has landed (and was sent)
        self.applied = [torch.cuda.Event() for _ in range(2)]    # receive buffers of set s have been consumed
        self.count = 0           # steps fed
        self.pending = None      # set index whose exchange is in flight / landed but not applied

    def _front(self, s, poses, verts_list, depth_list=None):
        torch, sh = self.torch, self.shard
        with torch.cuda.stream(self.gen_stream):
            if self.count >= 2:
                self.gen_stream.wait_event(self.ready[s])        # set s was last sent by exchange count-2
            sh.table.set_stream(self.gen_stream)
            sh.use_set(s)
            sh.generate_all(poses, verts_list, depth_list)
            self.generated[s].record(self.gen_stream)
        with torch.cuda.stream(self.front_stream):
            self.front_stream.wait_event(self.generated[s])
            if self.count >= 2:
                self.front_stream.wait_event(self.applied[s])    # set s was last applied as exchange count-2
            self.transport.all_to_all_bins(sh.bins_send, sh.bins_recv)
            self.transport.all_gather_packets(sh.packet.view(-1), sh.packets.view(sh.plan.world, -1))
            self.ready[s].record(self.front_stream)

    def _apply(self, s):
        torch, sh = self.torch, self.shard
        with torch.cuda.stream(self.table_stream):
            self.table_stream.wait_event(self.ready[s])
            sh.table.set_stream(self.table_stream)
            sh.use_set(s)
            sh.apply_all()
            self.applied[s].record(self.table_stream)

    def feed(self, poses, verts_list, depth_list=None):
        """Submit one step (batch frames of this rank's camera); applies the previous one."""
        s = self.count & 1
        self._front(s, poses, verts_list, depth_list)
        if self.pending is not None:
            self._apply(self.pending)
        self.pending = s
        self.count += 1

    def flush(self):
        if self.pending is not None:
            self._apply(self.pending)
            self.pending = None
        self.shard.table.set_stream(self.table_stream)
        self.table_stream.synchronize()
        self.front_stream.synchronize()
        self.gen_stream.synchronize()


def loopback_step(shards, poses, verts_list, depth_list=None):
    """The same step with every rank played in this process (no collective library):
    poses[r][b], verts_list[r][b] (depth_list[r][b] for sensor-depth packets)."""
    ex = LoopbackExchange(len(shards))
    for r, sh in enumerate(shards):
        sh.generate_all(poses[r], verts_list[r], None if depth_list is None else depth_list[r])
        ex.post(r, sh.bins_send, sh.packet)
    for r, sh in enumerate(shards):
        ex.fetch(r, sh.bins_recv, sh.packets)
        sh.apply_all()


def reference_multi_camera_frame(table, poses, verts_list):
    """What a sharded step must equal, on ONE unsharded table with step-level calls:
    one lock epoch, every camera's allocBlocks in camera order, then flatten + TSDF update
    camera by camera.  `table`: OracleTable (numpy verts) or SDFHashtable (device verts)."""
    table.reset_mutexes()
    for pose, verts in zip(poses, verts_list):
        table.set_pose(pose)
        table.alloc_blocks(verts)
    for pose, verts in zip(poses, verts_list):
        table.set_pose(pose)
        table.flatten()
        table.integrate_depth_map(verts)


def camera_phase(rank: int, world: int) -> float:
    """Cameras start evenly spread on the loop (C4: 90 degrees apart)."""
    return 2.0 * math.pi * rank / world


# ----------------------------------------------------------------------------
# bench (called by bench.py when WORLD_SIZE > 1)
# ----------------------------------------------------------------------------
def scaling_prediction(wl_name, world, batch):
    """The prediction written down BEFORE any multi-GPU run (profiles/r05_scaling_model.json, tools/scaling_model.py): frames/s at
    this world size from one-rank-at-a-time kernel times and bytes on the wire / a stated xGMI rate -- so that a measured curve
    is held against a number that was not fitted to it.  None where the model has no entry."""
    import json
    import os
    try:
        m = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r05_scaling_model.json")))
        e = m["workloads"][wl_name if wl_name in m["workloads"] else "C2"]["predicted"][str(world)]
        return dict(e, model="profiles/r05_scaling_model.json", batch_of_model=m["batch"],
                    note="written before any multi-GPU measurement; see the file for inputs and assumptions")
    except Exception:
        return None


def bench_sharded(args, wl, wl_name, rank, world, local_rank):
    """Timed windows of K exchanges on the bucket-range-sharded path (the process group is the caller's).
    Returns the result record on rank 0 and None elsewhere."""
    import json
    import os
    import statistics

    import torch
    import torch.distributed as dist

    from . import SEM_PINHOLE, default_params, synth

    Wd, Ht = wl["width"], wl["height"]
    nframes = min(args.frames or wl["frames"], 250)
    dev = torch.device("cuda", local_rank)
    # the few scalars the ranks agree on travel on the backend's own device (gloo: host; the test rig for
    # several ranks on one GPU)
    cdev = dev if dist.get_backend() == "nccl" else torch.device("cpu")
    stream = torch.cuda.Stream(device=dev)
    plan = ShardPlan(wl["buckets"], world)
    # Records per (camera, owner) key bin: one per 16 pixels, whatever the number of owners.  A
    # wave-deduplicated 640x480 room frame yields up to ~9 400 keys, and bucket-range ownership is skewed --
    # a camera facing a wall demands blocks whose hashes fall into few bucket ranges: with 8 owners a bin of
    # 4 800 records (twice the fullest bin of a sampled probe) still overflowed in the full run -- so a bin
    # must hold a whole frame's keys.  The bins travel at full capacity (19.7 MB per rank and exchange of
    # 8 frames at 8 ranks, half of what the depth packets weigh); a key that finds its bin full is counted
    # (key_bin_overflows) and demanded again by the next frame.
    capacity = max(2048, -(-Wd * Ht // 16))
    poses = synth.camera_loop(wl.get("loop", wl["frames"]), phase=camera_phase(rank, world))[:nframes]
    prims = synth.room_primitives()
    verts = torch.empty((nframes, Ht, Wd, 4), dtype=torch.float32, device=dev)
    # Sensor frames as the demo reads them (uint16 depth, 5000 units = 1 m, Application.cpp:38-42) and
    # the vertex maps preProcess makes from them: the packets then carry the 2-byte image instead of a
    # 4-byte camera-z plane (--float-packets: the unquantised float maps and float planes instead).
    sensor = not getattr(args, "float_packets", False) and (Wd * Ht) % 2 == 0
    from .hashtable import preprocess
    k_inv = np.linalg.inv(synth.K_matrix(Wd, Ht).astype(np.float64)).astype(np.float32)
    depth16 = torch.empty((nframes, Ht, Wd), dtype=torch.uint16, device=dev) if sensor else None
    scratch_n = torch.empty((Ht, Wd, 4), dtype=torch.float32, device=dev)
    for i in range(nframes):
        verts[i] = synth.render_room_verts(poses[i], Wd, Ht, prims, device=dev)
        if sensor:
            depth16[i] = (verts[i, :, :, 2] * 5000.0).round().clamp(0, 65535).to(torch.uint16)
            preprocess(depth16[i], k_inv, verts[i], scratch_n)
    torch.cuda.synchronize()
    params = default_params(numBuckets=wl["buckets"], numVoxelBlocks=wl["blocks"], voxelSize=wl["voxel"])
    transport = TorchDistTransport()
    batch = max(1, args.batch)
    pipelined = not getattr(args, "no_pipeline", False)
    # The exchange runs inside the library on RCCL directly (include/voxelhash_dist.h: one C call per exchange, three
    # HIP streams and three buffer sets in C++) unless --python-exchange asks for round 2's Python host (ShardedPipeline
    # over torch.distributed collectives), kept for comparison.
    # (the N > 1 test rig -- gloo, every rank on one GPU, which RCCL refuses -- goes through the Python host too)
    native = not getattr(args, "python_exchange", False) and pipelined and dist.get_backend() == "nccl"
    front = torch.cuda.Stream(device=dev)
    import ctypes as C
    native_error = None
    nd = None
    if native:
        # Every rank must take the same path, and must agree on it BEFORE any native collective: each rank first finds out
        # locally whether its library can bind RCCL (vh_dist_probe: no collective, nothing created), the outcomes are
        # reduced, and only then does rank 0 draw the id -- it always takes part in the broadcast -- and every rank call
        # ncclCommInitRank.  A failure after the agreement (out of memory while the shard is created, say) is fatal for
        # that rank: it exits and the launcher ends the others, instead of walking into a mismatched collective.
        from . import _lib as L
        probe = L.load().vh_dist_probe()
        if probe != 0:
            native_error = f"vh_dist_probe: {L.load().vh_last_error().decode()}"
        ok = torch.tensor([1 if probe == 0 else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            native = False
            native_error = native_error or "another rank's library could not bind RCCL"
        else:
            uid = unique_id(rank, torch_broadcast_bytes())
            nd = NativeDist(params, Wd, Ht, SEM_PINHOLE, rank, world, batch, uid, sensor_k_inv=k_inv if sensor else None,
                            key_capacity=0, device=local_rank)
            if world > 1:
                # before anything is fused or timed: a known pattern through ncclAllToAll / ncclAllGather, compared on the
                # device -- a transport that delivers to the wrong place fails here, loudly, on every rank that sees it
                nd.self_check()
    with torch.cuda.stream(stream):
        if native:
            # (key bins: the library's default -- ONE bin per (owner, batch) of 1.5 x batch x W*H/16 / world records)
            native_capacity = max(8192, (-(-Wd * Ht // 16) * batch * 3 // 2 + world - 1) // world + 1)

            nd.table.set_option("flatten_variant", 3)             # the line's value is the reference's walk (bench.py: Integrator); --option overrides
            for kv in getattr(args, "option", []) or []:          # A/B switches (bench.py --option name=value)
                k_, v_ = kv.split("=")
                if k_ in ("fused_generation", "force_collectives", "raycast_auto_start"):      # options of the exchange itself
                    nd.set_option(k_, int(v_))
                else:
                    nd.table.set_option(k_, int(v_))

            class _Shard:          # what the rest of this function reads of a HipShard
                table, packet_floats = nd.table, (36 + Wd * Ht // 2) if sensor else (32 + Wd * Ht)
            shard, pipe = _Shard, None
            # argument blocks of every start frame, prepared once (as the single-GPU Integrator does)
            src = depth16 if sensor else verts
            pose_blocks = [np.ascontiguousarray(np.stack([poses[(k + j) % nframes] for j in range(batch)]).reshape(batch, 16))
                           for k in range(nframes)]
            pose_ptrs = [a.ctypes.data_as(C.POINTER(C.c_float)) for a in pose_blocks]
            frame_ptrs = [(C.c_void_p * batch)(*[src[(k + j) % nframes].data_ptr() for j in range(batch)]) for k in range(nframes)]
        else:
            shard = HipShard(params, Wd, Ht, SEM_PINHOLE, plan, rank, capacity, batch=batch, device=dev, stream=stream,
                             sets=2 if pipelined else 1, sensor_k_inv=k_inv if sensor else None)
            shard.table.set_option("flatten_variant", 3)          # (as the native host above)
            for kv in getattr(args, "option", []) or []:
                k_, v_ = kv.split("=")
                if k_ not in ("fused_generation", "force_collectives", "raycast_auto_start"):
                    shard.table.set_option(k_, int(v_))
            pipe = ShardedPipeline(shard, transport, stream, front) if pipelined else None

        def step(i):
            if native:
                k = (i * batch) % nframes
                nd.step_raw(pose_ptrs[k], frame_ptrs[k])
                return
            ks = [(i * batch + b) % nframes for b in range(batch)]
            depths = [depth16[k] for k in ks] if sensor else None
            if pipe:
                pipe.feed([poses[k] for k in ks], [verts[k] for k in ks], depths)
            else:
                sharded_step(shard, transport, [poses[k] for k in ks], [verts[k] for k in ks], depths)

        def drain():
            if native:
                nd.flush()
            elif pipe:
                pipe.flush()
            shard.table.synchronize()
            torch.cuda.synchronize()

        # A full collection of the Python garbage collector walks every object torch has created
        # (~10^6): a 30-50 ms stall inside whatever call triggers it (seen in all_to_all_single), which
        # is several exchanges long.  Existing objects are moved out of its reach and it stays off
        # while the clock runs.
        import gc
        gc.collect()
        gc.freeze()
        gc.disable()
        # a fixed, untimed run-in before the W warm-up steps
        for i in range(100):
            step(i)
        drain()
        for i in range(args.warmup):
            step(i)
        drain()
        # windows of exactly K steps, each bracketed by a barrier + synchronisation on both sides, repeated
        # until ~0.3 s have been timed (every rank takes the same number: the decision is rank 0's)
        windows, nxt, host_enqueue = [], args.warmup, []
        while True:
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(args.steps):
                step(nxt + i)
            host_enqueue.append(time.perf_counter() - t0)     # (the host's share: enqueueing without waiting)
            drain()                  # every fed step has been applied when the clock stops
            dist.barrier()
            windows.append(time.perf_counter() - t0)
            nxt += args.steps
            go = torch.tensor([1 if (sum(windows) < 0.3 and len(windows) < 200) else 0], dtype=torch.int32, device=cdev)
            dist.broadcast(go, 0)
            if int(go.item()) == 0:
                break
        elapsed = statistics.median(windows)
        # per-dispatch HIP-event timing of the table kernels (separate, untimed pass)
        # Behind a flush the first two exchanges generate their keys with launches of their own and only the following ones carry
        # the generation in their frame launches (vh_dist option "fused_generation"): the steady-state launch is the difference
        # between a pass of 18 exchanges and a pass of 2.
        shard.table.set_profiling(True)
        for i in range(2):
            step(nxt + i)
        drain()
        kt2 = shard.table.kernel_times(reset=True)
        for i in range(18):
            step(nxt + 2 + i)
        drain()
        kt = shard.table.kernel_times(reset=True)
        for k_ in kt:
            if isinstance(kt[k_], (int, float)) and k_ in kt2:
                kt[k_] = kt[k_] - kt2[k_]
        shard.table.set_profiling(False)
        # per-exchange phase times from timing events of the library's own (separate, untimed pass): what a measured
        # scaling curve is read against -- the exchange's period is max(generate, collectives, apply) when the three streams
        # overlap as designed
        phases = None
        if native:
            nd.set_option("phase_timing", 1)
            for i in range(9):
                step(nxt + 3 + i)
            drain()
            ph = nd.phase_times()
            nd.set_option("phase_timing", 0)
            pt = torch.tensor([ph["generate"], ph["collectives"], ph["apply"], ph["first_to_last"], ph["host_enqueue"]],
                              dtype=torch.float64, device=cdev)
            dist.all_reduce(pt, op=dist.ReduceOp.MAX)
            phases = dict(zip(("generate", "collectives", "apply", "first_to_last", "host_enqueue"), [round(float(x), 2) for x in pt.tolist()]),
                          exchanges=ph["exchanges"], frames_per_camera_per_exchange=batch,
                          note="microseconds per exchange, the slowest rank's mean of each phase over 9 exchanges in a separate pass "
                               "with timing events (vh_dist_phase_times): key-generation launches (fused generation, the default: the frame launches "
                               "that carried it, i.e. the apply phase of an earlier exchange) / from the moment the collectives "
                               "may start to their completion (ncclAllToAll of the key bins + ncclAllGather of the packets; one rank: "
                               "nothing is sent) / the frame launches first to last / generation start to last frame launch / host "
                               "time inside vh_dist_step_batch")

        # raycast over the shards (extra, after the timed region).  With more than one rank it runs
        # only on request: the default multi-GPU run is the integration benchmark alone.
        do_raycast = world == 1 or getattr(args, "sharded_raycast", False)
        rc_elapsed, rc_iters, lost_total, view_cap, kte, ktv, view = 0.0, 20, 0, 8192, None, None, None
        rc_fixed_elapsed, lost_fixed, fixed_cap = 0.0, 0, 2048
        if do_raycast and native:
            # the raycast round inside the library (vh_dist_raycast): pose all-gather, export, ncclAllToAll of fixed record
            # slots, import, raycast on one stream, the pose by value -- no host synchronisation, no Python in between
            ray_depth = torch.empty((Ht, Wd), dtype=torch.float32, device=dev)
            lost_dev = torch.zeros(1, dtype=torch.int32, device=dev)
            for i in range(2):
                nd.raycast(poses[i], ray_depth, fixed_cap, lost=lost_dev)
            nd.flush()
            torch.cuda.synchronize()
            dist.barrier()
            t2 = time.perf_counter()
            for i in range(rc_iters):
                nd.raycast(poses[(7 * i) % nframes], ray_depth, fixed_cap, lost=lost_dev)
            nd.flush()
            torch.cuda.synchronize()
            dist.barrier()
            rc_fixed_elapsed = time.perf_counter() - t2
            lost_fixed = int(lost_dev.item())
            shard.table.set_profiling(True)
            for i in range(3):
                nd.raycast(poses[(7 * i) % nframes], ray_depth, fixed_cap)
            nd.flush()
            kte = shard.table.kernel_times(reset=True)
            shard.table.set_profiling(False)
        elif do_raycast:
            view = HipViewTable(params, Wd, Ht, SEM_PINHOLE, world, view_cap, device=dev, stream=stream)
            for i in range(2):
                sharded_raycast(shard, view, transport, poses[i], view_cap)
            torch.cuda.synchronize()
            dist.barrier()
            t1 = time.perf_counter()
            for i in range(rc_iters):
                _, lost = sharded_raycast(shard, view, transport, poses[(7 * i) % nframes], view_cap)
                lost_total += lost
            torch.cuda.synchronize()
            dist.barrier()
            rc_elapsed = time.perf_counter() - t1
            # the same without host synchronisation (fixed record slots, counts read on the device)
            fixed_cap, fstate = 2048, {}
            for i in range(2):
                sharded_raycast_fixed(shard, view, transport, poses[i], fixed_cap, state=fstate)
            torch.cuda.synchronize()
            dist.barrier()
            t2 = time.perf_counter()
            lost_dev = None
            for i in range(rc_iters):
                _, lost_f = sharded_raycast_fixed(shard, view, transport, poses[(7 * i) % nframes], fixed_cap, state=fstate)
                lost_dev = lost_f if lost_dev is None else lost_dev + lost_f
            torch.cuda.synchronize()
            dist.barrier()
            rc_fixed_elapsed = time.perf_counter() - t2
            lost_fixed = int(lost_dev.item())
            shard.table.set_profiling(True)
            view.table.set_profiling(True)
            for i in range(3):
                sharded_raycast(shard, view, transport, poses[(7 * i) % nframes], view_cap)
            torch.cuda.synchronize()
            kte, ktv = shard.table.kernel_times(reset=True), view.table.kernel_times(reset=True)
            shard.table.set_profiling(False)
            view.table.set_profiling(False)
    wt = torch.tensor(windows, dtype=torch.float64, device=cdev)
    dist.all_reduce(wt, op=dist.ReduceOp.MAX)                 # every window: the slowest rank's time
    windows = [float(x) for x in wt.tolist()]
    elapsed = statistics.median(windows)
    t = torch.tensor([rc_elapsed, rc_fixed_elapsed], dtype=torch.float64, device=cdev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    rc_elapsed, rc_fixed_elapsed = float(t[0].item()), float(t[1].item())
    c = shard.table.counters()
    stats = torch.tensor([c["occupied"], c["allocated_total"], c["bin_overflow"]], dtype=torch.int64, device=cdev)
    dist.all_reduce(stats, op=dist.ReduceOp.SUM)
    out = None
    comm_ranks = nd.comm_info()[1] if native else None
    forms = None
    if native:        # which form of the key generation each rank's exchanges ran in (a per-rank decision: the size rule reads the rank's own shard)
        ft = torch.zeros(world, dtype=torch.int32, device=cdev)
        ft[rank] = 1 if nd.generation_form() == "fused" else 0
        dist.all_reduce(ft, op=dist.ReduceOp.SUM)
        forms = ["fused" if int(x) else "separate" for x in ft.tolist()]
    if rank == 0:
        frames = args.steps * world * batch
        launches = max(1, kt["launches"])
        occ = c["occupied"]
        one_launch = kt.get("frame_pipelined_ms", 0) > 0
        indexed = False
        if one_launch:
            # the ONE launch of a multi-camera frame (frame_multi_pipelined_kernel: commit + TSDF update of frame b with the
            # claim + walk of frame b+1): one pass over rank 0's shard of the VoxelEntry array for all cameras (20 B per
            # owned entry), the compact entries + camera masks written (24 B), per visible block its entry and 4 KiB read +
            # 4 KiB written (cameras applied in registers)
            kname = "frame_multi_pipelined_kernel (rank 0)"
            walk_us = 1e3 * kt["frame_pipelined_ms"] / launches
            indexed = any(kv.replace(" ", "") == "flatten_variant=4" for kv in (getattr(args, "option", []) or []))
            if indexed:      # the walk-free multi-camera frame: bitmap + non-empty buckets in place of the 20*N walk
                kname = "frame_multi_pipelined_kernel (rank 0; walk role = flatten_index_tile_to<IndexSinkMulti>)"
                owned = shard.table.num_entries // 5
                walk_bytes = owned // 8 + 100 * (c["allocated_total"] - c.get("freed_total", 0)) + 24 * occ + occ * (20 + 4096 + 4096)
            else:
                walk_bytes = 20 * shard.table.num_entries + 24 * occ + occ * (20 + 4096 + 4096)
            pmc_key = "frame_multi_pipelined_kernel"
        else:
            # two launches per multi-camera frame: the dominant one = claim the bins || walk the shard for all cameras
            kname = "frame_multi_scan_claim_kernel (rank 0)"
            walk_us = 1e3 * kt["frame_scan_claim_ms"] / launches
            walk_bytes = 20 * shard.table.num_entries + 24 * occ
            pmc_key = "frame_multi_scan_claim_kernel"
        achieved = walk_bytes / (walk_us * 1e-6) / 1e9 if walk_us > 0 else 0.0
        # HBM traffic: rocprofv3 --pmc passes are single-process runs, so the counter figure is the one-rank
        # run's (profiles/pmc_latest.json, "<workload>sharded"): its measured bytes / algorithmic bytes ratio
        # applied to this rank's launch
        traffic, traffic_source = None, None
        try:
            pmc = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles",
                                              "pmc_latest.json"))).get(wl_name + "sharded", {})
            m, a = pmc.get(pmc_key + "_hbm_bytes_per_launch"), pmc.get("algorithmic_bytes_per_launch")
            if one_launch and indexed:
                m = None             # (no counter pass of the walk-free multi-camera launch exists: the reference walk's figure is not its)
            if m and a:
                traffic = int(round(walk_bytes * m / a)) if world > 1 else int(m)
                traffic_source = ("rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE of the one-rank run" if world == 1 else
                                  f"one-rank PMC ratio {m / a:.3f} x this rank's algorithmic bytes")
        except Exception:
            pass
        roofline = dict(bound="hbm" if not (one_launch and indexed) else "latency + VALU issue (no stream; the fraction is not a quality figure)", kernel=kname, achieved=round(achieved, 1),
                        peak=8000.0, unit="GB/s", frac=round(achieved / 8000.0, 4), traffic=traffic,
                        traffic_source=traffic_source,
                        bytes_per_launch=walk_bytes, us_per_launch=round(walk_us, 2),
                        launches_per_frame=1 if one_launch else 2,
                        commit_integrate_us=None if one_launch else round(1e3 * kt["frame_commit_integrate_ms"] / launches, 2))
        out = dict(
            metric=args.metric_name,
            value=round(frames / elapsed, 1), unit="frames/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
            ms_per_step=round(1e3 * elapsed / args.steps, 5), higher_is_better=True, scaling="weak",
            vs_baseline=None, dtype="f32", data="synthetic",
            windows=len(windows), timed_s=round(sum(windows), 4),
            host_enqueue_ms_per_step=round(1e3 * statistics.median(host_enqueue) / args.steps, 5),
            exchange_ranks=dict(dict(transport=(nd.transport if world > 1 else f"none (one rank: the frames are applied straight from the send buffers; "
                                                                                            f"{nd.transport} would carry them)"),
                                     ranks=comm_ranks, self_check="passed (vh_dist_self_check)" if world > 1 else "not run (one rank)")
                                if native else dict(transport="torch.distributed " + dist.get_backend(), ranks=dist.get_world_size()),
                                **({"shared_gpu": True} if os.environ.get("VH_BENCH_SHARE_GPU") == "1" and world > 1 else {})),
            exchange_phases_us=phases, predicted=scaling_prediction(wl_name, world, batch),
            generation_form=(forms[0] if forms and len(set(forms)) == 1 else forms),
            exchange_host=("libvoxelhash_hip.so: vh_dist_step_batch (include/voxelhash_dist.h)"
                           + (" on RCCL directly" if world > 1 else ", one rank: no collective")) if native
            else "Python: dist.ShardedPipeline over torch.distributed collectives (--python-exchange / --no-pipeline)"
                 + (f"; the native exchange was not available: {native_error}" if native_error else ""),
            config=dict(workload=f"{'C5' if wl_name == 'C5' else 'C4-style'}: {world} virtual {Wd}x{Ht} cameras (one per GPU) into one scene, "
                                 f"2^{int(math.log2(wl['buckets']))} buckets sharded by bucket range over {world} GPUs, "
                                 "RCCL all-to-all of block keys + all-gather of depth packets per step, PINHOLE; "
                                 + ("uint16 sensor depth, vertex maps by vh_preprocess, packets carry the uint16 image"
                                    if sensor else "float vertex maps, packets carry a float camera-z plane"),
                        frames_per_step=world * batch, frames_per_camera_per_exchange=batch,
                        resident_frames=nframes, key_bin_capacity=native_capacity if native else capacity,
                        key_bins="one per (owner, batch), records carry the frame (VH_BIN_PER_BATCH)" if native else "one per (owner, frame)",
                        key_bin_bytes_per_rank_and_exchange=16 * world * (native_capacity if native else capacity * batch),
                        pipelined=pipelined,
                        packet_bytes=4 * shard.packet_floats,
                        occupied_blocks_all_ranks=int(stats[0]), allocated_blocks_all_ranks=int(stats[1]),
                        key_bin_overflows=int(stats[2]), voxel_size=wl["voxel"],
                        voxel_blocks_per_rank=wl["blocks"]),
            roofline=roofline, cpu_baseline=None)
        if do_raycast and native:
            out["sharded_raycast"] = dict(
                mpix_per_s=round(world * rc_iters * Wd * Ht / rc_fixed_elapsed / 1e6, 1), views_per_round=world,
                ms_per_round=round(1e3 * rc_fixed_elapsed / rc_iters, 4), lost_records=lost_fixed,
                record_capacity_per_shard_and_view=fixed_cap, payload_bytes_per_rank=world * fixed_cap * VIEW_RECORD_BYTES,
                rank0_export_us=round(1e3 * kte["view_export_ms"] / 3, 2),
                note="vh_dist_raycast: every rank renders its own camera's view of the whole table inside the library -- "
                     "ncclAllGather of the poses, one walk of the shard for all views, ncclAllToAll of fixed slots of "
                     "{key, 512 voxels} records, import into a view table, raycast; one stream, no host synchronisation; "
                     "bit-equal to a raycast of the unsharded table")
        elif do_raycast:
            out["sharded_raycast"] = dict(
                mpix_per_s=round(world * rc_iters * Wd * Ht / rc_elapsed / 1e6, 1), views_per_round=world,
                ms_per_round=round(1e3 * rc_elapsed / rc_iters, 4), lost_records=lost_total,
                fixed_slots=dict(ms_per_round=round(1e3 * rc_fixed_elapsed / rc_iters, 4),
                                 mpix_per_s=round(world * rc_iters * Wd * Ht / rc_fixed_elapsed / 1e6, 1),
                                 record_capacity_per_shard_and_view=fixed_cap, lost_records=lost_fixed,
                                 payload_bytes_per_rank=world * fixed_cap * VIEW_RECORD_BYTES,
                                 note="sharded_raycast_fixed: poses gathered on the device, fixed record slots, "
                                      "counts read on the device -- no host synchronisation in the round"),
                record_capacity_per_shard_and_view=view_cap,
                rank0_export_us=round(1e3 * kte["view_export_ms"] / 3, 2),
                rank0_import_us=round(1e3 * ktv["view_import_ms"] / 3, 2),
                rank0_raycast_us=round(1e3 * ktv["raycast_ms"] / max(1, ktv["raycast_launches"]), 2),
                note="every rank renders its own camera's view of the whole table: one walk of its shard for "
                     "all views, all-to-all of {key, 512 voxels} records, import into a view table, raycast; "
                     "bit-equal to a raycast of the unsharded table")
    # orderly teardown while the streams the contexts were bound to are still alive (a context
    # destroyed by the garbage collector at interpreter exit synchronises a stream torch may
    # already have released)
    if view is not None:
        view.table.close()
    if native:
        nd.close()
    else:
        shard.table.close()
    torch.cuda.synchronize()
    dist.barrier()
    return out if rank == 0 else None
