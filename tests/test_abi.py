"""The drop-in boundary without a GPU: libvoxelhash_hip.so loads, exports every function
include/voxelhash.h declares, keeps the reference's record layouts, and fails loudly
(no CPU fallback) when no device is present."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "voxelhash.h")


DIST_HEADER = os.path.join(ROOT, "include", "voxelhash_dist.h")


def declared_functions(header=HEADER):
    src = open(header).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"^\s*(?:const\s+)?[A-Za-z_][\w\s\*]*?\b(\w+)\s*\([^;{]*\)\s*;", src, flags=re.M)
    return sorted(set(n for n in names if n not in ("defined",)))


def test_header_declares_the_reference_names():
    names = declared_functions()
    # VoxelUtils.h:5-13 (mapGLobjectsToCUDApointers intentionally absent: no GL interop)
    for n in ("updateConstantHashTableParams", "deviceAllocate", "deviceFree", "resetHashTableMutexes",
              "allocBlocks", "flattenIntoBuffer", "calculateKinectProjectionMatrix", "integrateDepthMap"):
        assert n in names
    for n in ("vh_create", "vh_integrate", "vh_raycast", "vh_generate_keys", "vh_insert_bins", "vh_integrate_packets"):
        assert n in names
    assert len(names) >= 35


def test_library_exports_every_declared_symbol(vh):
    from voxelhashing_demo_amd import _lib
    L = C.CDLL(_lib.LIB_PATH)
    missing = [n for n in declared_functions() if not hasattr(L, n)]
    assert not missing, f"declared in voxelhash.h but not exported: {missing}"
    # and the Python binding covers the same set
    assert sorted(_lib.SIGNATURES) == declared_functions()
    # the multi-GPU host (include/voxelhash_dist.h): exported, bound, and RCCL is NOT a link-time dependency
    dist = declared_functions(DIST_HEADER)
    assert "vh_dist_step_batch" in dist and "vh_dist_loopback_id" in dist and len(dist) == 18
    assert not [n for n in dist if not hasattr(L, n)]
    assert sorted(_lib.DIST_SIGNATURES) == dist
    import subprocess
    needed = subprocess.run(["readelf", "-d", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "rccl" not in needed.lower()


def test_record_layouts(vh):
    from voxelhashing_demo_amd import _lib
    assert C.sizeof(_lib.HashTableParams) == 176            # VoxelDataStructures.h:29-52
    assert vh.ENTRY_DTYPE.itemsize == 20                    # :20-26, pos@0 ptr@12 offset@16
    assert vh.ENTRY_DTYPE.fields["ptr"][1] == 12 and vh.ENTRY_DTYPE.fields["offset"][1] == 16
    assert vh.VOXEL_DTYPE.itemsize == 8                     # :12-17
    assert _lib.HashTableParams.numBuckets.offset == 128 and _lib.HashTableParams.integrationWeightMax.offset == 172


def test_default_params_are_common_h(vh):
    p = vh.default_params()
    assert (p.numBuckets, p.bucketSize, p.attachedLinkedListSize, p.numVoxelBlocks) == (5000, 5, 4, 1000)
    assert p.voxelBlockSize == 8 and p.numOccupiedBlocks == 0 and p.integrationWeightSample == 10
    assert p.voxelSize == pytest.approx(0.02) and p.truncation == 1.0 and p.integrationWeightMax == 255.0
    assert list(p.global_transform) == [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1]


def test_error_strings_and_argument_checks(vh):
    L = vh.load()
    assert L.vh_error_string(0) == b"ok"
    assert b"device" in L.vh_error_string(2)
    h = C.c_void_p()
    assert L.vh_create(None, C.byref(h)) == 1              # VH_ERR_INVALID_ARGUMENT
    assert L.vh_destroy(None) == 0
    assert L.vh_integrate(None, None, None, None) == 1
    assert L.vh_default_context() is None


def test_fails_loudly_without_a_gpu(vh):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    assert vh.load().vh_device_count() == 0
    with pytest.raises(vh.VoxelHashError, match="no usable HIP device"):
        vh.SDFHashtable(vh.default_params())


def test_facade_library_links(vh):
    from voxelhashing_demo_amd import _lib
    assert os.path.exists(_lib.FACADE_PATH)
    # C++ class of include/SDF_Hashtable.h: the mangled members must be there
    import subprocess
    syms = subprocess.run(["nm", "-DC", _lib.FACADE_PATH], capture_output=True, text=True).stdout
    for s in ("SDF_Hashtable::SDF_Hashtable()", "SDF_Hashtable::integrate(float4x4 const&, vh_float4 const*, vh_float4 const*)",
              "SDF_Hashtable::raycast(float4x4 const&, float*, float, float)", "SDF_Hashtable::~SDF_Hashtable()"):
        assert s in syms, s


def test_headers_compile_as_c99_and_cxx11(tmp_path):
    """The boundary is a plain C header (and two C++ facade headers): no torch, no HIP types."""
    import subprocess
    inc = os.path.join(ROOT, "include")
    c = tmp_path / "t.c"
    c.write_text('#include "voxelhash.h"\n#include "voxelhash_dist.h"\nint main(void) { return (int)sizeof(vh_icp_system) + (int)sizeof(vh_view_record) + (int)sizeof(vh_dist_config); }\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, "-fsyntax-only", str(c)], check=True)
    cpp = tmp_path / "t.cpp"
    cpp.write_text('#include "voxelhash.h"\n#include "SDF_Hashtable.h"\n#include "CameraTracking.h"\nint main() { return 0; }\n')
    subprocess.run(["g++", "-std=c++11", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, "-fsyntax-only", str(cpp)],
                   check=True)


def test_cpp_hosts_compile_against_the_headers():
    """The C++ programs the GPU tests build and run (tests/cpp/: one rank, threads over the loop-back transport, one process
    per rank over RCCL) still match include/SDF_Hashtable.h and include/voxelhash*.h -- checked here without a GPU."""
    import glob
    import subprocess
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc in this image")
    srcs = sorted(glob.glob(os.path.join(ROOT, "tests", "cpp", "*.cpp")))
    assert len(srcs) >= 3
    for src in srcs:
        subprocess.run([hipcc, "-std=c++17", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), src], check=True,
                       capture_output=True, timeout=300)
