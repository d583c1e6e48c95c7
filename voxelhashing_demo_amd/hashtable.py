"""Host-side mirror of the reference's SDF_Hashtable class (SDF_Hashtable.h:24-42)
for Python callers: same entry points (integrate, plus raycast standing in for
SDFRenderer::render), every call going straight through the C-ABI of
libvoxelhash_hip.so.  torch is used only as the owner of device buffers and of
the current stream; none of the path's arithmetic is done in torch.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L

ENTRY_DTYPE = np.dtype([("pos", "<i4", (3,)), ("ptr", "<i4"), ("offset", "<i4")])
VOXEL_DTYPE = np.dtype([("sdf", "<f4"), ("weight", "<f4")])


def default_params(**overrides) -> L.HashTableParams:
    """common.h:39-50 as copied by SDF_Hashtable.cpp:62-73, with overrides."""
    p = L.HashTableParams()
    L.load().vh_default_params(C.byref(p))
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def _dev_ptr(t) -> int:
    """Device address of a torch CUDA tensor (or a raw int address)."""
    if isinstance(t, int):
        return t
    if t is None:
        return 0
    if not t.is_cuda or not t.is_contiguous():
        raise ValueError("expected a contiguous CUDA tensor")
    return t.data_ptr()


def _pose16(pose):
    a = np.ascontiguousarray(np.asarray(pose, dtype=np.float32).reshape(16))
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def preprocess(depth_u16, k_inv, positions, normals, stream=None):
    """preProcess (CameraTrackingUtils.cu:115-120) on the GPU: uint16 depth [H, W] -> float4 vertex
    and normal maps [H, W, 4] (device tensors, written in place)."""
    H, W = depth_u16.shape
    k = np.ascontiguousarray(np.asarray(k_inv, np.float32).reshape(9))
    handle = 0 if stream is None else (stream if isinstance(stream, int) else stream.cuda_stream)
    L.check(L.load().vh_preprocess(_dev_ptr(depth_u16), k.ctypes.data_as(C.POINTER(C.c_float)), W, H,
                                   _dev_ptr(positions), _dev_ptr(normals), C.c_void_p(handle)), "vh_preprocess")
    return positions, normals


class SDFHashtable:
    """One voxel-hash table on one GPU.

    integrate(pose, verts, normals) == SDF_Hashtable::integrate (SDF_Hashtable.cpp:11-40);
    raycast(pose) stands in for SDFRenderer::render (SDFRenderer.cpp:210-255).
    """

    def __init__(self, params: L.HashTableParams | None = None, width: int = 640, height: int = 480,
                 semantics: int = L.SEM_REFERENCE, device: int = -1, bucket_range=None, stream=None):
        self._lib = L.load()
        self.params = params if params is not None else default_params()
        self.width, self.height, self.semantics = width, height, semantics
        cfg = L.Config(self.params, width, height, semantics, device)
        h = C.c_void_p()
        if bucket_range is None:
            L.check(self._lib.vh_create(C.byref(cfg), C.byref(h)), "vh_create")
            self.bucket_range = (0, self.params.numBuckets)
        else:
            lo, hi = bucket_range
            L.check(self._lib.vh_create_shard(C.byref(cfg), lo, hi, C.byref(h)), "vh_create_shard")
            self.bucket_range = (lo, hi)
        self._h = h
        self.stream_handle = 0                  # raw hipStream_t the context enqueues on (0 = default stream)
        if stream is not None:
            self.set_stream(stream)

    @classmethod
    def borrowed(cls, handle, params, width, height, semantics, bucket_range):
        """A view of a context somebody else owns (a vh_dist's shard): every query and option works, close() leaves it."""
        t = cls.__new__(cls)
        t._lib = L.load()
        t.params, t.width, t.height, t.semantics = params, width, height, semantics
        t.bucket_range = tuple(bucket_range)
        t._h = C.c_void_p(handle) if not isinstance(handle, C.c_void_p) else handle
        t._borrowed = True
        t.stream_handle = 0
        return t

    # ---- lifecycle ----
    def close(self):
        if getattr(self, "_h", None):
            if not getattr(self, "_borrowed", False):
                self._lib.vh_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream):
        """`stream`: a torch.cuda.Stream, or a raw hipStream_t address."""
        handle = stream if isinstance(stream, int) else stream.cuda_stream
        L.check(self._lib.vh_set_stream(self._h, C.c_void_p(handle)), "vh_set_stream")
        self.stream_handle = handle

    def set_projection(self, m):
        a = np.ascontiguousarray(np.asarray(m, np.float32).reshape(9))
        L.check(self._lib.vh_set_projection(self._h, a.ctypes.data_as(C.POINTER(C.c_float))), "vh_set_projection")

    def set_alloc_band(self, band_metres: float):
        """Opt-in truncation-band allocation; 0 = the reference's surface-block-only allocation."""
        L.check(self._lib.vh_set_alloc_band(self._h, float(band_metres)), "vh_set_alloc_band")

    def set_raycast_intrinsics(self, fx, fy, cx, cy):
        L.check(self._lib.vh_set_raycast_intrinsics(self._h, fx, fy, cx, cy), "vh_set_raycast_intrinsics")

    # ---- the hot path ----
    def integrate(self, pose, verts, normals=None):
        """Asynchronous: pose -> lock epoch -> allocBlocks -> flatten -> integrateDepthMap."""
        _, pp = _pose16(pose)
        L.check(self._lib.vh_integrate(self._h, pp, _dev_ptr(verts), _dev_ptr(normals)), "vh_integrate")

    def integrate_depth(self, pose, depth_u16, k_inv):
        """The frame straight from the uint16 sensor image [H, W] (== preprocess + integrate)."""
        _, pp = _pose16(pose)
        k = np.ascontiguousarray(np.asarray(k_inv, np.float32).reshape(9))
        L.check(self._lib.vh_integrate_depth(self._h, pp, _dev_ptr(depth_u16), k.ctypes.data_as(C.POINTER(C.c_float))),
                "vh_integrate_depth")

    def integrate_batch(self, poses, verts_list, normals_list=None):
        """len(poses) frames in len(poses) + 1 launches (pipelined frames, flushed at the end); equals
        integrate() frame by frame."""
        n = len(poses)
        p = np.ascontiguousarray(np.asarray(poses, np.float32).reshape(n, 16))
        v = (C.c_void_p * n)(*[_dev_ptr(t) for t in verts_list])
        nn = None if normals_list is None else (C.c_void_p * n)(*[_dev_ptr(t) for t in normals_list])
        L.check(self._lib.vh_integrate_batch(self._h, n, p.ctypes.data_as(C.POINTER(C.c_float)), v, nn), "vh_integrate_batch")

    def integrate_depth_batch(self, poses, depth_list, k_inv):
        n = len(poses)
        p = np.ascontiguousarray(np.asarray(poses, np.float32).reshape(n, 16))
        d = (C.c_void_p * n)(*[_dev_ptr(t) for t in depth_list])
        k = np.ascontiguousarray(np.asarray(k_inv, np.float32).reshape(9))
        L.check(self._lib.vh_integrate_depth_batch(self._h, n, p.ctypes.data_as(C.POINTER(C.c_float)), d,
                                                   k.ctypes.data_as(C.POINTER(C.c_float))), "vh_integrate_depth_batch")

    def flush(self):
        """Launch the pending half of the last pipelined frame (option "pipeline")."""
        L.check(self._lib.vh_flush(self._h), "vh_flush")

    def raycast(self, pose, out, t_min: float = 0.1, t_max: float = 5.0):
        _, pp = _pose16(pose)
        L.check(self._lib.vh_raycast(self._h, pp, t_min, t_max, _dev_ptr(out)), "vh_raycast")
        return out

    def raycast_normals(self, pose, out, normals, t_min: float = 0.1, t_max: float = 5.0):
        """The DDA raycast with the camera-frame normal map [H, W, 4] of the hits written by the same pass."""
        _, pp = _pose16(pose)
        L.check(self._lib.vh_raycast_normals(self._h, pp, t_min, t_max, _dev_ptr(out), _dev_ptr(normals)),
                "vh_raycast_normals")
        return out, normals

    def set_raycast_mode(self, mode: int):
        """RAYCAST_DDA (default) / RAYCAST_FIXED_STEP."""
        self.set_option("raycast_mode", int(mode))

    def raycast_maps(self, pose, depth, vertices, normals, t_min: float = 0.1, t_max: float = 5.0):
        """raycast + camera-frame vertex and normal maps of the same view (an ICP target)."""
        _, pp = _pose16(pose)
        L.check(self._lib.vh_raycast_maps(self._h, pp, t_min, t_max, _dev_ptr(depth), _dev_ptr(vertices),
                                          _dev_ptr(normals)), "vh_raycast_maps")
        return depth, vertices, normals

    def render_blocks(self, pose, front, back, t_min: float = 0.1, t_max: float = 5.0):
        """Block silhouettes (SDFRenderer::drawToFrontAndBack): nearest front / farthest back cube face per pixel."""
        _, pp = _pose16(pose)
        L.check(self._lib.vh_render_blocks(self._h, pp, t_min, t_max, _dev_ptr(front), _dev_ptr(back)), "vh_render_blocks")
        return front, back

    # ---- deletion / garbage collection ----
    def delete_blocks(self, keys, n: int = None):
        """keys: device int32 [n, 4] = {x, y, z, _}."""
        n = int(keys.shape[0]) if n is None else int(n)
        L.check(self._lib.vh_delete_blocks(self._h, _dev_ptr(keys), n), "vh_delete_blocks")

    def garbage_collect(self, sdf_threshold: float):
        L.check(self._lib.vh_garbage_collect(self._h, float(sdf_threshold)), "vh_garbage_collect")

    # ---- raycast over shards (DESIGN.md section 6 "raycast") ----
    VIEW_RECORD_BYTES = 4112

    def export_views(self, poses, records, capacity: int, counts, t_min: float = 0.1, t_max: float = 5.0):
        """poses: [n][16] host; records: device uint8 [n*capacity, 4112]; counts: device int32 [n]."""
        p = np.ascontiguousarray(np.asarray(poses, np.float32).reshape(-1, 16))
        L.check(self._lib.vh_export_views(self._h, p.ctypes.data_as(C.POINTER(C.c_float)), p.shape[0], t_min, t_max,
                                          _dev_ptr(records), capacity, _dev_ptr(counts)), "vh_export_views")

    def export_views_fixed(self, d_poses, n_views: int, records, capacity: int, counts, t_min: float = 0.1,
                           t_max: float = 5.0):
        """The same with device poses [n_views, 16] and fixed slots: view v's records at records[v*capacity:]."""
        L.check(self._lib.vh_export_views_fixed(self._h, _dev_ptr(d_poses), int(n_views), t_min, t_max, _dev_ptr(records),
                                                capacity, _dev_ptr(counts)), "vh_export_views_fixed")

    def import_views(self, records, num_sources: int, capacity: int, counts):
        """Fixed-slot import: source s holds min(counts[s], capacity) records at records[s*capacity:] (counts on the device)."""
        self._view_records = records
        L.check(self._lib.vh_import_views(self._h, _dev_ptr(records), int(num_sources), int(capacity), _dev_ptr(counts)),
                "vh_import_views")

    def import_view(self, records, count: int):
        """Make this dedicated, unsharded context hold exactly records[:count] (voxels stay in `records`)."""
        self._view_records = records            # keep the buffer alive while the table points into it
        L.check(self._lib.vh_import_view(self._h, _dev_ptr(records), int(count)), "vh_import_view")

    # ---- step-level entry points (VoxelUtils.h:5-13) ----
    def set_pose(self, pose):
        _, pp = _pose16(pose)
        L.check(self._lib.vh_set_pose(self._h, pp), "vh_set_pose")

    def reset_mutexes(self):
        L.check(self._lib.vh_reset_mutexes(self._h), "vh_reset_mutexes")

    def alloc_blocks(self, verts, normals=None):
        L.check(self._lib.vh_alloc_blocks(self._h, _dev_ptr(verts), _dev_ptr(normals)), "vh_alloc_blocks")

    def flatten(self, sync: bool = True):
        if not sync:
            L.check(self._lib.vh_flatten(self._h, None), "vh_flatten")
            return None
        n = C.c_int32()
        L.check(self._lib.vh_flatten(self._h, C.byref(n)), "vh_flatten")
        return n.value

    def integrate_depth_map(self, verts):
        L.check(self._lib.vh_integrate_depth_map(self._h, _dev_ptr(verts)), "vh_integrate_depth_map")

    # ---- sharding ----
    def generate_keys(self, verts, camera_id: int, num_shards: int, bins_out, capacity: int, packet_out=None,
                      bin_stride: int = 0):
        """Key bins (capacity records each, `bin_stride` records apart) + the camera packet for the
        pose set with set_pose().  bins_out / packet_out: device tensors or raw addresses."""
        L.check(self._lib.vh_generate_keys(self._h, _dev_ptr(verts), camera_id, num_shards, _dev_ptr(bins_out),
                                           capacity, bin_stride, _dev_ptr(packet_out)), "vh_generate_keys")

    def insert_bins(self, bins, num_bins: int, capacity: int, bin_stride: int = 0):
        L.check(self._lib.vh_insert_bins(self._h, _dev_ptr(bins), num_bins, capacity, bin_stride), "vh_insert_bins")

    def integrate_packets(self, num_cams: int, packets, packet_stride: int = 0):
        L.check(self._lib.vh_integrate_packets(self._h, num_cams, _dev_ptr(packets), packet_stride),
                "vh_integrate_packets")

    def generate_keys_batch(self, poses16, vert_ptrs, camera_id: int, num_shards: int, bins_out, capacity: int,
                            packets_out, batch: int, per_batch_bins: bool = False):
        """`batch` frames of this camera in one call.  poses16: float32 [batch, 16] (contiguous numpy),
        vert_ptrs: ctypes array of `batch` device addresses; bins_out [num_shards, batch, capacity, 4],
        packets_out [batch, 32 + W*H] (dense layouts).  per_batch_bins: bins_out [num_shards, capacity, 4], one bin per
        shard for the whole batch (VH_BIN_PER_BATCH)."""
        L.check(self._lib.vh_generate_keys_batch(
            self._h, batch, poses16.ctypes.data_as(C.POINTER(C.c_float)), vert_ptrs, camera_id, num_shards,
            _dev_ptr(bins_out), capacity, 0, -1 if per_batch_bins else 0, _dev_ptr(packets_out), 0), "vh_generate_keys_batch")

    def generate_keys_depth_batch(self, poses16, depth_ptrs, k_inv, camera_id: int, num_shards: int, bins_out,
                                  capacity: int, packets_out, batch: int, per_batch_bins: bool = False):
        """Keys + sensor-depth packets of `batch` frames from the uint16 images alone (dense layouts)."""
        k = np.ascontiguousarray(np.asarray(k_inv, np.float32).reshape(9))
        L.check(self._lib.vh_generate_keys_depth_batch(
            self._h, batch, poses16.ctypes.data_as(C.POINTER(C.c_float)), depth_ptrs,
            k.ctypes.data_as(C.POINTER(C.c_float)), camera_id, num_shards, _dev_ptr(bins_out), capacity, 0,
            -1 if per_batch_bins else 0, _dev_ptr(packets_out), 0), "vh_generate_keys_depth_batch")

    def write_packets_u16_batch(self, poses16, depth_ptrs, k_inv, packets_out, batch: int, packet_frame_stride: int = 0):
        """Sensor-depth packets (VH_PACKET_U16) of `batch` frames: depth_ptrs = ctypes array of device
        addresses of W*H uint16 images; packets_out: device tensor or address, [batch, 36 + W*H/2] floats."""
        k = np.ascontiguousarray(np.asarray(k_inv, np.float32).reshape(9))
        L.check(self._lib.vh_write_packets_u16_batch(
            self._h, batch, poses16.ctypes.data_as(C.POINTER(C.c_float)), depth_ptrs,
            k.ctypes.data_as(C.POINTER(C.c_float)), _dev_ptr(packets_out), packet_frame_stride),
            "vh_write_packets_u16_batch")

    def apply_frames_batch(self, bins, num_bins: int, capacity: int, num_cams: int, packets, batch: int,
                           per_batch_bins: bool = False):
        """Apply `batch` multi-camera frames: bins [num_bins, batch, capacity, 4] (per_batch_bins: [num_bins, capacity, 4]),
        packets [num_cams, batch, 32 + W*H] (dense layouts)."""
        L.check(self._lib.vh_apply_frames_batch(self._h, batch, _dev_ptr(bins), num_bins, capacity, 0,
                                                -1 if per_batch_bins else 0, num_cams, _dev_ptr(packets), 0, 0),
                "vh_apply_frames_batch")

    # ---- model dump / checkpoint ----
    def dump_sdf_text(self, path: str):
        """SDF_dump.txt in the format of SDFRenderer::printSDFdata (SDFRenderer.cpp:71-110)."""
        L.check(self._lib.vh_dump_sdf_text(self._h, str(path).encode()), "vh_dump_sdf_text")

    def save_snapshot(self, path: str):
        L.check(self._lib.vh_save_snapshot(self._h, str(path).encode()), "vh_save_snapshot")

    def load_snapshot(self, path: str):
        L.check(self._lib.vh_load_snapshot(self._h, str(path).encode()), "vh_load_snapshot")

    # ---- queries ----
    def synchronize(self):
        L.check(self._lib.vh_synchronize(self._h), "vh_synchronize")

    def counters(self) -> dict:
        c = L.Counters()
        L.check(self._lib.vh_get_counters(self._h, C.byref(c)), "vh_get_counters")
        return c.as_dict()

    def device_pointers(self) -> L.PtrContainer:
        p = L.PtrContainer()
        L.check(self._lib.vh_get_device_pointers(self._h, C.byref(p)), "vh_get_device_pointers")
        return p

    @property
    def num_entries(self) -> int:
        lo, hi = self.bucket_range
        return (hi - lo) * self.params.bucketSize

    def _download(self, which, dtype, count):
        out = np.empty(count, dtype)
        if count:
            L.check(self._lib.vh_download(self._h, which, out.ctypes.data_as(C.c_void_p), out.nbytes), "vh_download")
        return out

    def hash_table(self) -> np.ndarray:
        return self._download(L.BUF_HASH_TABLE, ENTRY_DTYPE, self.num_entries)

    def compact(self) -> np.ndarray:
        n = self.counters()["occupied"]
        return self._download(L.BUF_COMPACT, ENTRY_DTYPE, n)

    def sdf_blocks(self) -> np.ndarray:
        return self._download(L.BUF_SDF_BLOCKS, VOXEL_DTYPE, self.params.numVoxelBlocks * 512)

    def heap(self) -> np.ndarray:
        return self._download(L.BUF_HEAP, np.dtype("<u4"), self.params.numVoxelBlocks)

    def block_voxels(self, ptr: int) -> np.ndarray:
        """The 512 voxels of the block at voxel index `ptr` (an entry's ptr), without the rest of the volume."""
        out = np.empty(512, VOXEL_DTYPE)
        L.check(self._lib.vh_download_range(self._h, L.BUF_SDF_BLOCKS, 8 * int(ptr), out.ctypes.data_as(C.c_void_p),
                                            out.nbytes), "vh_download_range")
        return out

    def allocated(self) -> np.ndarray:
        t = self.hash_table()
        return t[t["ptr"] != L.FREE_BLOCK]

    def debug_eval(self, points, out):
        L.check(self._lib.vh_debug_eval(self._h, _dev_ptr(points), points.shape[0], _dev_ptr(out)), "vh_debug_eval")

    def set_option(self, name: str, value: int):
        L.check(self._lib.vh_set_option(self._h, name.encode(), int(value)), "vh_set_option")

    def set_profiling(self, on: bool):
        L.check(self._lib.vh_set_profiling(self._h, int(on)), "vh_set_profiling")

    def kernel_times(self, reset: bool = True) -> dict:
        t = L.KernelTimes()
        L.check(self._lib.vh_get_kernel_times(self._h, C.byref(t), int(reset)), "vh_get_kernel_times")
        return t.as_dict()
