"""ctypes binding of the CPU oracle (oracle/vh_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (voxelhashing_demo_amd) never
imports this module.  Parity status: see the header of vh_oracle.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libvh_oracle.so")

SEM_REFERENCE = 0
SEM_PINHOLE = 1
FREE_BLOCK = -1
BAND_RAY, BAND_NORMAL_DDA, BAND_RAY_DDA = 0, 1, 2
INT_DEPTH_TRUNCATION, INT_WEIGHT_SAMPLE = 1, 2
RAYCAST_FIXED_STEP, RAYCAST_DDA = 0, 1
POS_SENTINEL = 0x7FFFFFFF

ENTRY_DTYPE = np.dtype([("pos", "<i4", (3,)), ("ptr", "<i4"), ("offset", "<i4")])
VOXEL_DTYPE = np.dtype([("sdf", "<f4"), ("weight", "<f4")])
assert ENTRY_DTYPE.itemsize == 20 and VOXEL_DTYPE.itemsize == 8


class Params(C.Structure):
    """HashTableParams, VoxelDataStructures.h:29-52 (176 bytes)."""
    _fields_ = [
        ("global_transform", C.c_float * 16),
        ("inv_global_transform", C.c_float * 16),
        ("numBuckets", C.c_uint32),
        ("bucketSize", C.c_uint32),
        ("attachedLinkedListSize", C.c_uint32),
        ("numVoxelBlocks", C.c_uint32),
        ("voxelBlockSize", C.c_int32),
        ("voxelSize", C.c_float),
        ("numOccupiedBlocks", C.c_uint32),
        ("maxIntegrationDistance", C.c_float),
        ("truncScale", C.c_float),
        ("truncation", C.c_float),
        ("integrationWeightSample", C.c_uint32),
        ("integrationWeightMax", C.c_float),
    ]


assert C.sizeof(Params) == 176


class FrameStats(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in (
        "pixels_valid", "pixels_in_frustum", "inserted", "lock_losses", "bucket_full",
        "heap_exhausted", "occupied", "voxels_updated")] + [("heap_counter", C.c_int32)]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (seconds)."""
    srcs = [os.path.join(_HERE, n) for n in ("vh_oracle.c", "vh_icp_oracle.c", "vh_oracle.h")]
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(f) for f in srcs):
        subprocess.run(["make", "-C", _HERE, "-B"], check=True, capture_output=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        fp = C.POINTER(C.c_float)
        ip = C.POINTER(C.c_int32)
        L.vho_default_params.argtypes = [C.POINTER(Params)]
        L.vho_create.restype = C.c_void_p
        L.vho_create.argtypes = [C.POINTER(Params), C.c_int, C.c_int, C.c_int]
        L.vho_destroy.argtypes = [C.c_void_p]
        L.vho_set_projection.argtypes = [C.c_void_p, fp]
        L.vho_set_raycast_intrinsics.argtypes = [C.c_void_p] + [C.c_float] * 4
        L.vho_set_alloc_band.argtypes = [C.c_void_p, C.c_float]
        L.vho_set_band_mode.argtypes = [C.c_void_p, C.c_int]
        L.vho_set_normals.argtypes = [C.c_void_p, fp]
        L.vho_set_overflow.argtypes = [C.c_void_p, C.c_int, C.c_uint32]
        L.vho_set_integrate_flags.argtypes = [C.c_void_p, C.c_int]
        L.vho_set_pose.argtypes = [C.c_void_p, fp]
        L.vho_reset_mutexes.argtypes = [C.c_void_p]
        L.vho_alloc_blocks.argtypes = [C.c_void_p, fp]
        L.vho_flatten.argtypes = [C.c_void_p]
        L.vho_flatten.restype = C.c_int
        L.vho_integrate_depth_map.argtypes = [C.c_void_p, fp]
        L.vho_integrate.argtypes = [C.c_void_p, fp, fp, C.POINTER(FrameStats)]
        L.vho_integrate.restype = C.c_int
        L.vho_render_blocks.argtypes = [C.c_void_p, fp, C.c_float, C.c_float, fp, fp]
        L.vho_integrate_mt.argtypes = [C.c_void_p, fp, fp, C.c_int, C.POINTER(FrameStats)]
        L.vho_integrate_mt.restype = C.c_int
        L.vho_raycast.argtypes = [C.c_void_p, fp, C.c_float, C.c_float, fp]
        L.vho_raycast_dda.argtypes = [C.c_void_p, fp, C.c_float, C.c_float, C.c_int, fp, fp]
        L.vho_raycast_dda.restype = None
        L.vho_get_params.restype = C.POINTER(Params)
        L.vho_get_params.argtypes = [C.c_void_p]
        for name in ("vho_hash_table", "vho_compact_table", "vho_sdf_blocks", "vho_heap"):
            getattr(L, name).restype = C.c_void_p
            getattr(L, name).argtypes = [C.c_void_p]
        L.vho_compact_count.argtypes = [C.c_void_p]
        L.vho_compact_count.restype = C.c_int
        L.vho_heap_counter.argtypes = [C.c_void_p]
        L.vho_heap_counter.restype = C.c_int
        L.vho_float2int_rz.argtypes = [C.c_float]
        L.vho_float2int_rz.restype = C.c_int32
        L.vho_hash.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_uint32]
        L.vho_hash.restype = C.c_uint32
        L.vho_world2voxel.argtypes = [fp, C.c_float, ip]
        L.vho_voxel2block.argtypes = [ip, C.c_int32, ip]
        L.vho_world2block.argtypes = [fp, C.c_float, C.c_int32, ip]
        L.vho_invert4x4.argtypes = [fp, fp]
        L.vho_mat4_mul_vec4.argtypes = [fp, fp, fp]
        L.vho_project.argtypes = [fp, fp, ip]
        L.vho_block_in_frustum.argtypes = [C.c_void_p, ip]
        L.vho_block_in_frustum.restype = C.c_int
        L.vho_launch_rank.argtypes = [C.c_int, C.c_int, C.c_int]
        L.vho_launch_rank.restype = C.c_uint32
        L.vho_create_shard.restype = C.c_void_p
        L.vho_create_shard.argtypes = [C.POINTER(Params), C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint32]
        L.vho_generate_keys.argtypes = [C.c_void_p, fp, C.c_uint32, C.c_int, ip, C.c_int]
        L.vho_generate_keys.restype = C.c_int
        L.vho_insert_bins.argtypes = [C.c_void_p, ip, C.c_int, C.c_int]
        L.vho_insert_bins.restype = C.c_int
        L.vho_write_packet.argtypes = [C.c_void_p, fp, fp]
        L.vho_integrate_packets.argtypes = [C.c_void_p, C.c_int, fp]
        L.vho_integrate_packets.restype = C.c_int
        L.vho_preprocess.argtypes = [C.POINTER(C.c_uint16), fp, C.c_int, C.c_int, fp, fp]
        dp = C.POINTER(C.c_double)
        L.vho_depth_to_maps.argtypes = [fp, fp, C.c_int, C.c_int, fp, fp]
        L.vho_icp_build_system.argtypes = [fp, fp, fp, fp, fp, C.c_float, C.c_int, C.c_int, C.c_int, dp, dp, dp,
                                           C.POINTER(C.c_uint32)]
        L.vho_icp_correspondences.argtypes = [fp, fp, fp, fp, fp, C.c_float, C.c_int, C.c_int, C.c_int, fp, fp, fp,
                                              C.POINTER(C.c_uint32)]
        L.vho_icp_correspondences.restype = C.c_double
        L.vho_se3_exp.argtypes = [dp, dp]
        L.vho_se3_log.argtypes = [dp, dp]
        L.vho_icp_solve.argtypes = [dp, dp, dp]
        L.vho_icp_solve.restype = C.c_int
        L.vho_icp_align.argtypes = [fp, fp, fp, fp, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, fp, dp,
                                    C.POINTER(C.c_uint32)]
        L.vho_icp_align.restype = C.c_int
        L.vho_delete_blocks.argtypes = [C.c_void_p, ip, C.c_int]
        L.vho_delete_blocks.restype = C.c_int
        L.vho_garbage_collect.argtypes = [C.c_void_p, C.c_float]
        L.vho_garbage_collect.restype = C.c_int
        L.vho_export_view.argtypes = [C.c_void_p, fp, C.c_float, C.c_float, C.c_void_p, C.c_int]
        L.vho_export_view.restype = C.c_int
        L.vho_import_view.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.vho_import_view.restype = C.c_int
        L.vho_view_frustum.argtypes = [C.c_void_p, fp, C.c_float, C.c_float, fp]
        L.vho_view_holds_block.argtypes = [C.c_void_p, fp, ip]
        L.vho_view_holds_block.restype = C.c_int
        _lib = L
    return _lib


def _fptr(a: np.ndarray):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_float))


def default_params(**overrides) -> Params:
    p = Params()
    lib().vho_default_params(C.byref(p))
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


# ---- scalar helpers (known-answer tests) ----
def float2int_rz(x: float) -> int:
    return int(lib().vho_float2int_rz(float(x)))


def hash_block(x: int, y: int, z: int, num_buckets: int) -> int:
    return int(lib().vho_hash(x, y, z, num_buckets))


def world2voxel(p, voxel_size: float):
    a = np.asarray(p, np.float32).copy()
    out = (C.c_int32 * 3)()
    lib().vho_world2voxel(_fptr(a), voxel_size, out)
    return tuple(out)


def voxel2block(v, block_size: int = 8):
    a = (C.c_int32 * 3)(*[int(c) for c in v])
    out = (C.c_int32 * 3)()
    lib().vho_voxel2block(a, block_size, out)
    return tuple(out)


def world2block(p, voxel_size: float, block_size: int = 8):
    a = np.asarray(p, np.float32).copy()
    out = (C.c_int32 * 3)()
    lib().vho_world2block(_fptr(a), voxel_size, block_size, out)
    return tuple(out)


def invert4x4(m) -> np.ndarray:
    a = np.ascontiguousarray(np.asarray(m, np.float32).reshape(16))
    out = np.empty(16, np.float32)
    lib().vho_invert4x4(_fptr(a), _fptr(out))
    return out.reshape(4, 4)


def project(m, p):
    a = np.ascontiguousarray(np.asarray(m, np.float32).reshape(9))
    b = np.asarray(p, np.float32).copy()
    out = (C.c_int32 * 2)()
    lib().vho_project(_fptr(a), _fptr(b), out)
    return tuple(out)


def launch_rank(x: int, y: int, width: int) -> int:
    return int(lib().vho_launch_rank(x, y, width))


def preprocess(depth_u16, k_inv):
    """(positions, normals) float32 [H, W, 4] from a uint16 depth image (preProcess,
    CameraTrackingUtils.cu:115-120)."""
    d = np.ascontiguousarray(depth_u16, np.uint16)
    H, W = d.shape
    k = np.ascontiguousarray(np.asarray(k_inv, np.float32).reshape(9))
    pos = np.empty((H, W, 4), np.float32)
    nrm = np.empty((H, W, 4), np.float32)
    lib().vho_preprocess(d.ctypes.data_as(C.POINTER(C.c_uint16)), _fptr(k), W, H, _fptr(pos), _fptr(nrm))
    return pos, nrm


class OracleTable:
    """Scalar CPU mirror of SDF_Hashtable (SDF_Hashtable.h:24-42)."""

    def __init__(self, params: Params | None = None, width: int = 640, height: int = 480,
                 semantics: int = SEM_REFERENCE, bucket_range=None):
        self.params = params if params is not None else default_params()
        self.width, self.height, self.semantics = width, height, semantics
        self.bucket_range = tuple(bucket_range) if bucket_range else (0, self.params.numBuckets)
        self._h = lib().vho_create_shard(C.byref(self.params), width, height, semantics, *self.bucket_range)
        if not self._h:
            raise MemoryError("vho_create failed")
        self.last_stats = None

    # ---- step-level entry points ----
    def set_pose(self, pose):
        pose = np.ascontiguousarray(np.asarray(pose, np.float32).reshape(16))
        lib().vho_set_pose(self._h, _fptr(pose))

    def reset_mutexes(self):
        lib().vho_reset_mutexes(self._h)

    def alloc_blocks(self, verts):
        lib().vho_alloc_blocks(self._h, _fptr(np.ascontiguousarray(verts, np.float32)))

    def flatten(self) -> int:
        return int(lib().vho_flatten(self._h))

    def integrate_depth_map(self, verts):
        lib().vho_integrate_depth_map(self._h, _fptr(np.ascontiguousarray(verts, np.float32)))

    # ---- bucket-range sharding (build extension) ----
    def packet_floats(self) -> int:
        return 32 + self.width * self.height

    def generate_keys(self, verts, camera_id: int, num_shards: int, capacity: int):
        """(bins [num_shards, capacity, 4] int32, packet float32) for the pose set with set_pose()."""
        verts = np.ascontiguousarray(verts, np.float32)
        bins = np.zeros((num_shards, capacity, 4), np.int32)
        worst = lib().vho_generate_keys(self._h, _fptr(verts), camera_id, num_shards,
                                        bins.ctypes.data_as(C.POINTER(C.c_int32)), capacity)
        if worst > capacity - 1:
            raise OverflowError(f"key bin overflow: {worst} keys, capacity {capacity - 1}")
        packet = np.empty(self.packet_floats(), np.float32)
        lib().vho_write_packet(self._h, _fptr(verts), _fptr(packet))
        return bins, packet

    def insert_bins(self, bins) -> int:
        bins = np.ascontiguousarray(bins, np.int32)
        return int(lib().vho_insert_bins(self._h, bins.ctypes.data_as(C.POINTER(C.c_int32)), bins.shape[0],
                                         bins.shape[1]))

    def integrate_packets(self, packets) -> int:
        packets = np.ascontiguousarray(packets, np.float32).reshape(-1, self.packet_floats())
        return int(lib().vho_integrate_packets(self._h, packets.shape[0], _fptr(packets)))

    def close(self):
        if getattr(self, "_h", None):
            lib().vho_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:       # interpreter shutdown: module globals may already be gone
            pass

    def set_projection(self, m):
        a = np.ascontiguousarray(np.asarray(m, np.float32).reshape(9))
        lib().vho_set_projection(self._h, _fptr(a))

    def set_raycast_intrinsics(self, fx, fy, cx, cy):
        lib().vho_set_raycast_intrinsics(self._h, fx, fy, cx, cy)

    def set_alloc_band(self, band_metres: float, mode: int = None):
        lib().vho_set_alloc_band(self._h, float(band_metres))
        if mode is not None:
            lib().vho_set_band_mode(self._h, int(mode))

    def set_normals(self, normals):
        """Normal map [H, W, 4] (camera frame) of the frames that follow, for BAND_NORMAL_DDA; None = none."""
        self._normals = None if normals is None else np.ascontiguousarray(normals, np.float32)
        lib().vho_set_normals(self._h, None if normals is None else _fptr(self._normals))

    def set_overflow(self, enabled: bool, segment_buckets: int = 0):
        """Overflow linked list on / off; chains wrap inside segments of `segment_buckets` buckets (0: own range)."""
        lib().vho_set_overflow(self._h, int(bool(enabled)), int(segment_buckets))

    def set_integrate_flags(self, flags: int):
        lib().vho_set_integrate_flags(self._h, int(flags))

    def integrate(self, pose, verts, normals=None) -> int:
        pose = np.ascontiguousarray(np.asarray(pose, np.float32).reshape(16))
        verts = np.ascontiguousarray(verts, np.float32)
        assert verts.size == self.width * self.height * 4
        if normals is not None:
            self.set_normals(normals)
        st = FrameStats()
        occ = lib().vho_integrate(self._h, _fptr(pose), _fptr(verts), C.byref(st))
        self.last_stats = st.as_dict()
        return int(occ)

    def integrate_mt(self, pose, verts, threads: int) -> int:
        """integrate() on `threads` host threads (identical results)."""
        pose = np.ascontiguousarray(np.asarray(pose, np.float32).reshape(16))
        verts = np.ascontiguousarray(verts, np.float32)
        st = FrameStats()
        occ = lib().vho_integrate_mt(self._h, _fptr(pose), _fptr(verts), int(threads), C.byref(st))
        self.last_stats = st.as_dict()
        return int(occ)

    def set_raycast_mode(self, mode: int):
        """RAYCAST_DDA (default: the voxel DDA of raycastSDF.frag:121-177) or RAYCAST_FIXED_STEP (rounds 1-2)."""
        self.raycast_mode = int(mode)

    def raycast(self, pose, t_min: float = 0.1, t_max: float = 5.0, jumps: bool = True, normals: bool = False):
        """Depth image (and, with normals=True, the camera-frame normal map [H, W, 4]) in the table's raycast
        mode.  jumps=False walks absent blocks voxel by voxel (same bits; the defining form of the DDA)."""
        pose = np.ascontiguousarray(np.asarray(pose, np.float32).reshape(16))
        out = np.zeros((self.height, self.width), np.float32)
        if getattr(self, "raycast_mode", RAYCAST_DDA) == RAYCAST_FIXED_STEP:
            if normals:
                raise ValueError("the fixed-step march has no normal output")
            lib().vho_raycast(self._h, _fptr(pose), t_min, t_max, _fptr(out))
            return out
        nrm = np.zeros((self.height, self.width, 4), np.float32) if normals else None
        lib().vho_raycast_dda(self._h, _fptr(pose), t_min, t_max, 1 if jumps else 0, _fptr(out),
                              _fptr(nrm) if normals else None)
        return (out, nrm) if normals else out

    # ---- deletion / garbage collection ----
    def delete_blocks(self, keys) -> int:
        k = np.zeros((len(keys), 4), np.int32)
        if len(keys):
            k[:, :3] = np.asarray(keys, np.int32).reshape(-1, 3)
        return int(lib().vho_delete_blocks(self._h, k.ctypes.data_as(C.POINTER(C.c_int32)), len(keys)))

    def garbage_collect(self, sdf_threshold: float) -> int:
        return int(lib().vho_garbage_collect(self._h, float(sdf_threshold)))

    # ---- raycast over shards: view records are uint8 [count, 4112] = {pos[3], 0, 512 voxels} ----
    VIEW_RECORD_BYTES = 4112

    def export_view(self, pose, capacity: int, t_min: float = 0.1, t_max: float = 5.0):
        """(records[min(count, capacity)], count): this table's allocated blocks the view can touch."""
        pose = np.ascontiguousarray(np.asarray(pose, np.float32).reshape(16))
        rec = np.zeros((max(1, capacity), self.VIEW_RECORD_BYTES), np.uint8)
        n = int(lib().vho_export_view(self._h, _fptr(pose), t_min, t_max, rec.ctypes.data, capacity))
        return rec[:min(n, capacity)], n

    def view_frustum(self, pose, t_min: float = 0.1, t_max: float = 5.0) -> np.ndarray:
        pose = np.ascontiguousarray(np.asarray(pose, np.float32).reshape(16))
        f = np.zeros(22, np.float32)
        lib().vho_view_frustum(self._h, _fptr(pose), t_min, t_max, _fptr(f))
        return f

    def view_holds_block(self, frustum, block) -> bool:
        a = (C.c_int32 * 3)(*[int(c) for c in block])
        return bool(lib().vho_view_holds_block(self._h, _fptr(frustum), a))

    def import_view(self, records) -> int:
        """Make this (otherwise unused, unsharded) table hold exactly `records`; returns the drops."""
        self._view_records = np.ascontiguousarray(records, np.uint8)      # the voxels stay in here
        n = self._view_records.size // self.VIEW_RECORD_BYTES
        return int(lib().vho_import_view(self._h, self._view_records.ctypes.data, n))

    def render_blocks(self, pose, t_min: float = 0.1, t_max: float = 5.0):
        """(front, back) [H, W]: camera depth of the nearest front / farthest back face of the allocated blocks' cubes."""
        pose = np.ascontiguousarray(np.asarray(pose, np.float32).reshape(16))
        front, back = np.zeros((self.height, self.width), np.float32), np.zeros((self.height, self.width), np.float32)
        lib().vho_render_blocks(self._h, _fptr(pose), t_min, t_max, _fptr(front), _fptr(back))
        return front, back

    def block_in_frustum(self, block) -> bool:
        a = (C.c_int32 * 3)(*[int(c) for c in block])
        return bool(lib().vho_block_in_frustum(self._h, a))

    # ---- views into the oracle's memory (copy before the table is destroyed) ----
    def _n_entries(self):
        lo, hi = self.bucket_range
        return (hi - lo) * self.params.bucketSize

    def hash_table(self) -> np.ndarray:
        n = self._n_entries()
        buf = (C.c_char * (n * 20)).from_address(lib().vho_hash_table(self._h))
        return np.frombuffer(buf, dtype=ENTRY_DTYPE, count=n)

    def compact(self) -> np.ndarray:
        n = int(lib().vho_compact_count(self._h))
        if n == 0:
            return np.zeros(0, ENTRY_DTYPE)
        buf = (C.c_char * (n * 20)).from_address(lib().vho_compact_table(self._h))
        return np.frombuffer(buf, dtype=ENTRY_DTYPE, count=n)

    def sdf_blocks(self) -> np.ndarray:
        n = self.params.numVoxelBlocks * 512
        buf = (C.c_char * (n * 8)).from_address(lib().vho_sdf_blocks(self._h))
        return np.frombuffer(buf, dtype=VOXEL_DTYPE, count=n)

    def heap_counter(self) -> int:
        return int(lib().vho_heap_counter(self._h))

    def heap(self) -> np.ndarray:
        n = self.params.numVoxelBlocks
        buf = (C.c_char * (n * 4)).from_address(lib().vho_heap(self._h))
        return np.frombuffer(buf, dtype=np.uint32, count=n)

    def compact_count(self) -> int:
        return int(lib().vho_compact_count(self._h))

    def allocated(self) -> np.ndarray:
        t = self.hash_table()
        return t[t["ptr"] != FREE_BLOCK]

    def block_voxels(self, entry) -> np.ndarray:
        ptr = int(entry["ptr"])
        return self.sdf_blocks()[ptr:ptr + 512]


# ---- frame-to-frame ICP (vh_icp_oracle.c) ----
ICP_ABS_DISTANCE, ICP_NEED_TARGET = 1, 2


def _f32(a, n=None):
    a = np.ascontiguousarray(np.asarray(a, np.float32))
    assert n is None or a.size == n
    return a


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def depth_to_maps(depth, k_inv):
    """float depth [H, W] in metres -> (positions [H, W, 4], normals [H, W, 4])."""
    depth = _f32(depth)
    H, W = depth.shape
    k = _f32(k_inv, 9)
    pos, nrm = np.zeros((H, W, 4), np.float32), np.zeros((H, W, 4), np.float32)
    lib().vho_depth_to_maps(_fptr(depth), _fptr(k), W, H, _fptr(pos), _fptr(nrm))
    return pos, nrm


def icp_build_system(inp, target, normals, delta, K, dist_thres, flags=0):
    inp, target, normals = _f32(inp), _f32(target), _f32(normals)
    H, W = inp.shape[:2]
    JTJ, JTr, err, cnt = np.zeros(36), np.zeros(6), C.c_double(), C.c_uint32()
    lib().vho_icp_build_system(_fptr(inp), _fptr(target), _fptr(normals), _fptr(_f32(delta, 16)), _fptr(_f32(K, 9)),
                               dist_thres, W, H, flags, _dptr(JTJ), _dptr(JTr), C.byref(err), C.byref(cnt))
    return JTJ.reshape(6, 6), JTr, err.value, cnt.value


def icp_correspondences(inp, target, normals, delta, K, dist_thres, flags=0):
    """-> (corres [H,W,4], corres_normals [H,W,4], residuals [H,W], error, count)"""
    inp, target, normals = _f32(inp), _f32(target), _f32(normals)
    H, W = inp.shape[:2]
    c, cn, r = np.empty((H, W, 4), np.float32), np.empty((H, W, 4), np.float32), np.empty((H, W), np.float32)
    cnt = C.c_uint32()
    err = lib().vho_icp_correspondences(_fptr(inp), _fptr(target), _fptr(normals), _fptr(_f32(delta, 16)),
                                        _fptr(_f32(K, 9)), dist_thres, W, H, flags, _fptr(c), _fptr(cn), _fptr(r),
                                        C.byref(cnt))
    return c, cn, r, float(err), cnt.value


def se3_exp(twist):
    t, T = np.ascontiguousarray(twist, np.float64), np.zeros(16)
    lib().vho_se3_exp(_dptr(t), _dptr(T))
    return T.reshape(4, 4)


def se3_log(T):
    T, t = np.ascontiguousarray(np.asarray(T, np.float64).reshape(16)), np.zeros(6)
    lib().vho_se3_log(_dptr(T), _dptr(t))
    return t


def icp_solve(JTJ, JTr, estimate):
    est = np.ascontiguousarray(estimate, np.float64).copy()
    ok = lib().vho_icp_solve(_dptr(np.ascontiguousarray(JTJ, np.float64).reshape(36)),
                             _dptr(np.ascontiguousarray(JTr, np.float64)), _dptr(est))
    return bool(ok), est


def icp_align(inp, target, normals, K, dist_thres=0.08, max_iters=20, flags=0, delta=None):
    """-> (delta [4,4] float32, iterations, last error, last count)."""
    inp, target, normals = _f32(inp), _f32(target), _f32(normals)
    H, W = inp.shape[:2]
    d = _f32(np.eye(4) if delta is None else delta, 16).reshape(16).copy()
    err, cnt = C.c_double(), C.c_uint32()
    it = lib().vho_icp_align(_fptr(inp), _fptr(target), _fptr(normals), _fptr(_f32(K, 9)), dist_thres, W, H,
                             max_iters, flags, _fptr(d), C.byref(err), C.byref(cnt))
    return d.reshape(4, 4), int(it), err.value, cnt.value
