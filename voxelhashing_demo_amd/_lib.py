"""ctypes loader for libvoxelhash_hip.so (the C-ABI declared in include/voxelhash.h).

There is no CPU fallback: if the HIP library is missing the import of anything
that needs it raises, and on a machine without a GPU `vh_create` returns
VH_ERR_NO_DEVICE which is surfaced as an exception.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VOXELHASH_LIB") or os.path.join(_HERE, "lib", "libvoxelhash_hip.so")   # override: tuning builds
FACADE_PATH = os.path.join(_HERE, "lib", "libsdf_hashtable.so")

VH_OK = 0
SEM_REFERENCE = 0
SEM_PINHOLE = 1
BUF_HASH_TABLE, BUF_COMPACT, BUF_SDF_BLOCKS, BUF_HEAP = 0, 1, 2, 3
FREE_BLOCK = -1


class HashTableParams(C.Structure):
    """VoxelDataStructures.h:29-52 (176 bytes)."""
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


class Config(C.Structure):
    _fields_ = [("params", HashTableParams), ("width", C.c_int32), ("height", C.c_int32),
                ("semantics", C.c_int32), ("device", C.c_int32)]


class Counters(C.Structure):
    _fields_ = [("occupied", C.c_int32), ("heap_counter", C.c_int32), ("allocated_total", C.c_uint32),
                ("heap_exhausted", C.c_uint32), ("candidates", C.c_uint32), ("epoch", C.c_uint32),
                ("bin_overflow", C.c_uint32), ("freed_total", C.c_uint32), ("last_freed", C.c_uint32),
                ("cand_overflow", C.c_uint32), ("spin_timeouts", C.c_uint32)]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class KernelTimes(C.Structure):
    _fields_ = [("launches", C.c_uint64), ("alloc_claim_ms", C.c_double), ("alloc_commit_ms", C.c_double),
                ("flatten_ms", C.c_double), ("integrate_ms", C.c_double), ("raycast_ms", C.c_double),
                ("raycast_launches", C.c_uint64), ("frame_scan_claim_ms", C.c_double),
                ("frame_commit_integrate_ms", C.c_double), ("view_export_ms", C.c_double),
                ("view_import_ms", C.c_double), ("gc_ms", C.c_double), ("gc_calls", C.c_uint64),
                ("render_blocks_ms", C.c_double), ("frame_pipelined_ms", C.c_double)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class IcpSystem(C.Structure):
    _fields_ = [("JTJ", C.c_double * 36), ("JTr", C.c_double * 6), ("error", C.c_double), ("count", C.c_uint32)]


class PtrContainer(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "d_heap", "d_hashTable", "d_compactifiedHashTable", "d_hashTableBucketMutex", "d_SDFBlocks",
        "d_heapCounter", "d_compactifiedHashCounter")]


assert C.sizeof(HashTableParams) == 176

# every symbol include/voxelhash.h declares: (restype, argtypes)
_vp, _i32, _u32, _f = C.c_void_p, C.c_int32, C.c_uint32, C.c_float
_fp = C.POINTER(C.c_float)
SIGNATURES = {
    "vh_default_params": (None, [C.POINTER(HashTableParams)]),
    "vh_error_string": (C.c_char_p, [C.c_int]),
    "vh_last_error": (C.c_char_p, []),
    "vh_device_count": (C.c_int, []),
    "vh_create": (C.c_int, [C.POINTER(Config), C.POINTER(_vp)]),
    "vh_destroy": (C.c_int, [_vp]),
    "vh_set_stream": (C.c_int, [_vp, _vp]),
    "vh_set_projection": (C.c_int, [_vp, _fp]),
    "vh_set_raycast_intrinsics": (C.c_int, [_vp, _f, _f, _f, _f]),
    "vh_set_alloc_band": (C.c_int, [_vp, _f]),
    "vh_set_pose": (C.c_int, [_vp, _fp]),
    "vh_reset_mutexes": (C.c_int, [_vp]),
    "vh_alloc_blocks": (C.c_int, [_vp, _vp, _vp]),
    "vh_flatten": (C.c_int, [_vp, C.POINTER(_i32)]),
    "vh_integrate_depth_map": (C.c_int, [_vp, _vp]),
    "vh_integrate": (C.c_int, [_vp, _fp, _vp, _vp]),
    "vh_integrate_depth": (C.c_int, [_vp, _fp, _vp, _fp]),
    "vh_raycast": (C.c_int, [_vp, _fp, _f, _f, _vp]),
    "vh_raycast_normals": (C.c_int, [_vp, _fp, _f, _f, _vp, _vp]),
    "vh_debug_set_raycast_stamps": (C.c_int, [_vp, _vp]),
    "vh_debug_occupy": (C.c_int, [_vp, _vp, _i32, _i32]),
    "vh_render_blocks": (C.c_int, [_vp, _fp, _f, _f, _vp, _vp]),
    "vh_icp_create": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "vh_icp_destroy": (C.c_int, [_vp]),
    "vh_icp_set_stream": (C.c_int, [_vp, _vp]),
    "vh_icp_build_system": (C.c_int, [_vp, _vp, _vp, _vp, _fp, _fp, _f, C.c_int32, C.POINTER(IcpSystem)]),
    "vh_icp_correspondences": (C.c_int, [_vp, _vp, _vp, _vp, _fp, _fp, _f, C.c_int32, _vp, _vp, _vp,
                                         C.POINTER(IcpSystem)]),
    "vh_icp_solve": (C.c_int, [C.POINTER(IcpSystem), C.POINTER(C.c_double)]),
    "vh_se3_exp": (None, [C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "vh_se3_log": (None, [C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "vh_icp_align": (C.c_int, [_vp, _vp, _vp, _vp, _fp, _f, C.c_int32, C.c_int32, _fp, C.POINTER(IcpSystem),
                               C.POINTER(C.c_int32)]),
    "vh_raycast_maps": (C.c_int, [_vp, _fp, _f, _f, _vp, _vp, _vp]),
    "vh_fusion_step": (C.c_int, [_vp, _vp, _vp, _fp, _fp, _f, C.c_int32, C.c_int32, _f, _f, _vp, _vp, _vp, _vp, _vp,
                                 C.POINTER(C.c_double), C.POINTER(IcpSystem), C.POINTER(C.c_int32)]),
    "vh_depth_to_maps": (C.c_int, [_vp, _fp, C.c_int32, C.c_int32, _vp, _vp, _vp]),
    "computeCorrespondences": (C.c_float, [_vp, _vp, _vp, _vp, _vp, _vp, _fp, C.c_int, C.c_int]),
    "vh_generate_keys_depth_batch": (C.c_int, [_vp, C.c_int32, _fp, C.POINTER(_vp), _fp, C.c_uint32, C.c_int32, _vp,
                                               C.c_int32, C.c_int32, C.c_int32, _vp, C.c_size_t]),
    "vh_write_packets_u16_batch": (C.c_int, [_vp, C.c_int32, _fp, C.POINTER(_vp), _fp, _vp, C.c_size_t]),
    "vh_delete_blocks": (C.c_int, [_vp, _vp, C.c_int32]),
    "vh_garbage_collect": (C.c_int, [_vp, _f]),
    "vh_export_views": (C.c_int, [_vp, _fp, C.c_int32, _f, _f, _vp, C.c_int32, _vp]),
    "vh_import_view": (C.c_int, [_vp, _vp, C.c_int32]),
    "vh_synchronize": (C.c_int, [_vp]),
    "vh_get_counters": (C.c_int, [_vp, C.POINTER(Counters)]),
    "vh_get_params": (C.c_int, [_vp, C.POINTER(HashTableParams)]),
    "vh_get_device_pointers": (C.c_int, [_vp, C.POINTER(PtrContainer)]),
    "vh_download": (C.c_int, [_vp, C.c_int, _vp, C.c_size_t]),
    "vh_download_range": (C.c_int, [_vp, C.c_int, C.c_size_t, _vp, C.c_size_t]),
    "vh_export_views_fixed": (C.c_int, [_vp, _vp, C.c_int32, C.c_float, C.c_float, _vp, C.c_int32, _vp]),
    "vh_import_views": (C.c_int, [_vp, _vp, C.c_int32, C.c_int32, _vp]),
    "vh_flush": (C.c_int, [_vp]),
    "vh_integrate_batch": (C.c_int, [_vp, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "vh_integrate_depth_batch": (C.c_int, [_vp, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_void_p), C.POINTER(C.c_float)]),
    "vh_debug_eval": (C.c_int, [_vp, _vp, _i32, _vp]),
    "vh_set_option": (C.c_int, [_vp, C.c_char_p, C.c_int]),
    "vh_set_profiling": (C.c_int, [_vp, C.c_int]),
    "vh_get_kernel_times": (C.c_int, [_vp, C.POINTER(KernelTimes), C.c_int]),
    "vh_create_shard": (C.c_int, [C.POINTER(Config), _u32, _u32, C.POINTER(_vp)]),
    "vh_generate_keys": (C.c_int, [_vp, _vp, _u32, _i32, _vp, _i32, _i32, _vp]),
    "vh_insert_bins": (C.c_int, [_vp, _vp, _i32, _i32, _i32]),
    "vh_integrate_packets": (C.c_int, [_vp, _i32, _vp, C.c_size_t]),
    "vh_generate_keys_batch": (C.c_int, [_vp, _i32, _fp, C.POINTER(_vp), _u32, _i32, _vp, _i32, _i32, _i32, _vp,
                                         C.c_size_t]),
    "vh_apply_frames_batch": (C.c_int, [_vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp, C.c_size_t, C.c_size_t]),
    "vh_dump_sdf_text": (C.c_int, [_vp, C.c_char_p]),
    "vh_save_snapshot": (C.c_int, [_vp, C.c_char_p]),
    "vh_load_snapshot": (C.c_int, [_vp, C.c_char_p]),
    "vh_preprocess": (C.c_int, [_vp, _fp, _i32, _i32, _vp, _vp, _vp]),
    "SetCameraIntrinsic": (C.c_bool, [_fp, _fp]),
    "preProcess": (None, [_vp, _vp, _vp]),
    "updateConstantHashTableParams": (None, [C.POINTER(HashTableParams)]),
    "deviceAllocate": (None, [C.POINTER(HashTableParams)]),
    "deviceFree": (None, []),
    "resetHashTableMutexes": (None, [C.POINTER(HashTableParams)]),
    "allocBlocks": (None, [_vp, _vp]),
    "flattenIntoBuffer": (C.c_int, [C.POINTER(HashTableParams)]),
    "calculateKinectProjectionMatrix": (None, []),
    "integrateDepthMap": (None, [C.POINTER(HashTableParams), _vp]),
    "vh_default_context": (_vp, []),
}



class DistPhases(C.Structure):
    """vh_dist_phases (include/voxelhash_dist.h): sums over the exchanges completed since the last reset."""
    _fields_ = [("generate_us", C.c_double), ("collectives_us", C.c_double), ("apply_us", C.c_double),
                ("first_to_last_us", C.c_double), ("host_enqueue_us", C.c_double), ("exchanges", C.c_uint64)]


class DistConfig(C.Structure):
    """vh_dist_config of include/voxelhash_dist.h."""
    _fields_ = [("table", Config), ("rank", C.c_int32), ("world", C.c_int32), ("batch", C.c_int32),
                ("key_capacity", C.c_int32), ("packet_format", C.c_int32), ("k_inv", C.c_float * 9)]


# every symbol include/voxelhash_dist.h declares (the multi-GPU host on RCCL)
DIST_SIGNATURES = {
    "vh_dist_unique_id": (C.c_int, [C.c_char_p]),
    "vh_dist_create": (C.c_int, [C.POINTER(DistConfig), C.c_char_p, _vp, C.POINTER(_vp)]),
    "vh_dist_destroy": (C.c_int, [_vp]),
    "vh_dist_shard": (_vp, [_vp]),
    "vh_dist_step_batch": (C.c_int, [_vp, _fp, C.POINTER(_vp)]),
    "vh_dist_flush": (C.c_int, [_vp]),
    "vh_dist_raycast": (C.c_int, [_vp, _fp, _f, _f, C.c_int32, _vp, _vp]),
    "vh_dist_host_stats": (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "vh_dist_set_option": (C.c_int, [_vp, C.c_char_p, C.c_int32]),
    "vh_dist_phase_times": (C.c_int, [_vp, C.POINTER(DistPhases), C.c_int32]),
    "vh_dist_self_check": (C.c_int, [_vp]),
    "vh_dist_probe": (C.c_int, []),
    "vh_dist_raycast_auto": (C.c_int, [_vp, _fp, _f, _f, _vp, _vp, C.POINTER(C.c_int32)]),
    "vh_dist_comm_info": (C.c_int, [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "vh_dist_loopback_id": (C.c_int, [C.c_char_p]),
    "vh_dist_transport_name": (C.c_char_p, [_vp]),
    "vh_dist_generation_form": (C.c_int, [_vp]),
    "vh_dist_set_user_stream": (C.c_int, [_vp, _vp, C.c_int32]),
}

_lib = None


class VoxelHashError(RuntimeError):
    pass


def load():
    """Load the HIP library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VoxelHashError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7.  Importing
        # torch first puts that copy in the process, and the loader then binds this library's
        # NEEDED libamdhip64.so.7 to it (same SONAME) instead of opening /opt/rocm's second copy,
        # which would fail to see the GPU.  Without torch the RUNPATH copy (/opt/rocm) is used.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in list(SIGNATURES.items()) + list(DIST_SIGNATURES.items()):
            if name in DIST_SIGNATURES and not hasattr(L, name):
                continue                   # (a tuning build of an older commit, tools/ab_commits.sh)
            fn = getattr(L, name)          # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc: int, where: str):
    if rc != VH_OK:
        L = load()
        raise VoxelHashError(f"{where}: {L.vh_error_string(rc).decode()} ({L.vh_last_error().decode()})")
