// vh_api.hip -- the C-ABI of libvoxelhash_hip.so (include/voxelhash.h): context
// lifecycle, per-frame orchestration (SDF_Hashtable.cpp:11-40 without its four
// device syncs and two D2H reads), and the reference's drop-in names.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "vh_kernels.hip"

using namespace vh;

// ---------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------
static thread_local std::string g_last_error;

static int fail(int code, const char *what, hipError_t e = hipSuccess)
{
    g_last_error = what;
    if (e != hipSuccess) {
        g_last_error += ": ";
        g_last_error += hipGetErrorString(e);
    }
    return code;
}

#define VH_HIP(call)                                                              \
    do {                                                                          \
        hipError_t e_ = (call);                                                   \
        if (e_ != hipSuccess)                                                     \
            return fail(e_ == hipErrorOutOfMemory ? VH_ERR_OUT_OF_MEMORY : VH_ERR_HIP, #call, e_); \
    } while (0)

// ---------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------
// Profiling attaches a start/stop event pair to each dispatch (hipExtLaunchKernelGGL):
// the pair carries the begin/end timestamps of that kernel alone, the same
// quantity rocprofv3 --kernel-trace reports, without the inter-kernel gaps a
// hipEventRecord pair would add.
enum Phase : int {
    kPhaseClaim = 0, kPhaseCommit, kPhaseFlatten, kPhaseIntegrate, kPhaseRaycast,
    kPhaseFrameScanClaim, kPhaseFrameCommitIntegrate, kPhaseViewExport, kPhaseViewImport, kPhaseGc, kPhaseRaycastBounds, kNumPhases
};

struct TimedLaunch {
    int phase;
    hipEvent_t start, stop;
};

struct vh_context {
    HashTableParams params;
    FrameParams fp;
    DevPtrs dp;
    int device = 0;
    hipStream_t stream = nullptr;
    size_t numEntries = 0;         // owned entries
    uint32_t ownedBuckets = 0;
    float rc_fx = 0, rc_fy = 0, rc_cx = 0, rc_cy = 0;
    bool profiling = false;
    std::vector<TimedLaunch> timed;
    uint64_t profiledFrames = 0;
    vh_kernel_times times{};
    int integrateGrid = 2048;
    int persistentBlocks = 2048;   // workgroups of the persistent walk (flatten_variant 5)
    int raycastPatch = 1;          // pixels of a raycast wave: 1 = 8x8 square, 0 = 16x4 rows
    int raycastXcd = 1;            // tiles renumbered so that each XCD (own L2) renders a contiguous part of the image
    int packetFormat = VH_PACKET_F32;   // what vh_integrate_packets / vh_apply_frames_batch read
    int fusedFrame = 1;            // vh_integrate as two launches (0: the four step kernels)
    int commitBlocks = 128;        // workgroups serving candidates in the fused second launch
    int fusedParity = 0;           // which of the two per-frame counter sets the next fused frame uses
    bool compactArmed = false;     // alloc_commit has zeroed the compact counter and no flatten ran since
    // WalkKind.  Same-process A/B of the fused frame (launch 1 + launch 2, us): C2  3: 17.2 + 5.1,
    // 6 (mask form): 16.8 + 13.6, 1: 17.8, 2: 18.9, 5: 17.7, 4 (index, not the reference walk): 6.4 + 5.3;
    // C3  3: 88 + 22, 6: 74 + 51, 4: 43 + 19.  The mask form makes launch 1 a pure stream but launch 2
    // then serialises mask -> entry -> atomic -> block update inside each workgroup.
    int flattenVariant = 3;
    int occupiedCounter = kCompactCount;   // which device counter holds the occupied count of the last frame
    // raycast over shards
    int32_t *viewLists = nullptr;          // export: selected entry indices, [views][capacity]
    size_t viewListsSize = 0;              // in int32
    int32_t *blockList = nullptr;          // vh_render_blocks: counter (4 ints) + indices of all allocated entries
    const Voxel *viewBlocks = nullptr;     // import: the record buffer the view table's ptrs address
    int32_t viewCount = 0;                 // records of the last import (their buckets are listed in compactMask)
};

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
        if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
    }
    ~DeviceGuard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

// ---------------------------------------------------------------------------
// host math: float4x4::getInverse, cuda_SimpleMatrixUtil.h:944-1069.  The table
// lists, per output element, the six signed triple products in the order the
// reference sums them, so the fp32 result has the same bits (this translation
// unit is built with -ffp-contract=off).
// ---------------------------------------------------------------------------
static const signed char k_cof[16][6][4] = {
    {{+1,5,10,15},{-1,5,11,14},{-1,9,6,15},{+1,9,7,14},{+1,13,6,11},{-1,13,7,10}},
    {{-1,1,10,15},{+1,1,11,14},{+1,9,2,15},{-1,9,3,14},{-1,13,2,11},{+1,13,3,10}},
    {{+1,1,6,15},{-1,1,7,14},{-1,5,2,15},{+1,5,3,14},{+1,13,2,7},{-1,13,3,6}},
    {{-1,1,6,11},{+1,1,7,10},{+1,5,2,11},{-1,5,3,10},{-1,9,2,7},{+1,9,3,6}},
    {{-1,4,10,15},{+1,4,11,14},{+1,8,6,15},{-1,8,7,14},{-1,12,6,11},{+1,12,7,10}},
    {{+1,0,10,15},{-1,0,11,14},{-1,8,2,15},{+1,8,3,14},{+1,12,2,11},{-1,12,3,10}},
    {{-1,0,6,15},{+1,0,7,14},{+1,4,2,15},{-1,4,3,14},{-1,12,2,7},{+1,12,3,6}},
    {{+1,0,6,11},{-1,0,7,10},{-1,4,2,11},{+1,4,3,10},{+1,8,2,7},{-1,8,3,6}},
    {{+1,4,9,15},{-1,4,11,13},{-1,8,5,15},{+1,8,7,13},{+1,12,5,11},{-1,12,7,9}},
    {{-1,0,9,15},{+1,0,11,13},{+1,8,1,15},{-1,8,3,13},{-1,12,1,11},{+1,12,3,9}},
    {{+1,0,5,15},{-1,0,7,13},{-1,4,1,15},{+1,4,3,13},{+1,12,1,7},{-1,12,3,5}},
    {{-1,0,5,11},{+1,0,7,9},{+1,4,1,11},{-1,4,3,9},{-1,8,1,7},{+1,8,3,5}},
    {{-1,4,9,14},{+1,4,10,13},{+1,8,5,14},{-1,8,6,13},{-1,12,5,10},{+1,12,6,9}},
    {{+1,0,9,14},{-1,0,10,13},{-1,8,1,14},{+1,8,2,13},{+1,12,1,10},{-1,12,2,9}},
    {{-1,0,5,14},{+1,0,6,13},{+1,4,1,14},{-1,4,2,13},{-1,12,1,6},{+1,12,2,5}},
    {{+1,0,5,10},{-1,0,6,9},{-1,4,1,10},{+1,4,2,9},{+1,8,1,6},{-1,8,2,5}},
};

static void invert4x4(const float e[16], float out[16])
{
    float inv[16];
    for (int o = 0; o < 16; ++o) {
        float acc = 0.0f;
        for (int k = 0; k < 6; ++k) {
            const signed char *c = k_cof[o][k];
            float t = (e[c[1]] * e[c[2]]) * e[c[3]];
            if (c[0] < 0) t = -t;
            acc = (k == 0) ? t : acc + t;
        }
        inv[o] = acc;
    }
    const float det = e[0] * inv[0] + e[1] * inv[4] + e[2] * inv[8] + e[3] * inv[12];
    const float detr = 1.0f / det;
    for (int i = 0; i < 16; ++i) out[i] = inv[i] * detr;
}

// ---------------------------------------------------------------------------
// small exported helpers
// ---------------------------------------------------------------------------
extern "C" void vh_default_params(HashTableParams *p)
{
    static const float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    std::memset(p, 0, sizeof *p);
    std::memcpy(p->global_transform, I, sizeof I);
    std::memcpy(p->inv_global_transform, I, sizeof I);
    p->numBuckets = 5000;              // common.h:39-50
    p->bucketSize = 5;
    p->attachedLinkedListSize = 4;
    p->numVoxelBlocks = 1000;
    p->voxelBlockSize = 8;
    p->voxelSize = 0.02f;
    p->numOccupiedBlocks = 0;
    p->maxIntegrationDistance = 4.0f;
    p->truncScale = 0.01f;
    p->truncation = 1.0f;
    p->integrationWeightSample = 10;
    p->integrationWeightMax = 255.0f;
}

extern "C" const char *vh_error_string(int code)
{
    switch (code) {
        case VH_OK: return "ok";
        case VH_ERR_INVALID_ARGUMENT: return "invalid argument";
        case VH_ERR_NO_DEVICE: return "no usable HIP device";
        case VH_ERR_OUT_OF_MEMORY: return "out of device memory";
        case VH_ERR_HIP: return "HIP runtime error";
        case VH_ERR_NOT_INITIALISED: return "deviceAllocate() has not been called";
        case VH_ERR_SINGULAR: return "singular linear system";
        default: return "unknown error";
    }
}

extern "C" const char *vh_last_error(void) { return g_last_error.c_str(); }

extern "C" int vh_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ---------------------------------------------------------------------------
// lifecycle
// ---------------------------------------------------------------------------
static void default_projection(vh_context *c)
{
    // common.h:7-10 scaled with the resolution; REFERENCE keeps the transposed
    // matrix the reference actually uploads (common.h:16 read row-major by
    // cuda_SimpleMatrixUtil.h:316-320 at VoxelUtils.cu:227)
    const float sx = (float)c->fp.width / 640.0f, sy = (float)c->fp.height / 480.0f;
    const float fx = 517.3f * sx, fy = 516.5f * sy, cx = 318.6f * sx, cy = 255.3f * sy;
    const float KT[9] = {fx, 0, 0, 0, fy, 0, cx, cy, 1};
    const float K[9] = {fx, 0, cx, 0, fy, cy, 0, 0, 1};
    std::memcpy(c->fp.proj, c->fp.semantics == VH_SEM_REFERENCE ? KT : K, sizeof K);
    c->rc_fx = fx; c->rc_fy = fy; c->rc_cx = cx; c->rc_cy = cy;
}

static int free_buffers(vh_context *c)
{
    if (c->dp.heap) (void)hipFree(c->dp.heap);
    if (c->dp.table) (void)hipFree(c->dp.table);
    if (c->dp.compact) (void)hipFree(c->dp.compact);
    if (c->dp.claim) (void)hipFree(c->dp.claim);
    if (c->dp.blocks) (void)hipFree(c->dp.blocks);
    if (c->dp.counters) (void)hipFree(c->dp.counters);
    if (c->dp.candidates) (void)hipFree(c->dp.candidates);
    if (c->dp.compactMask) (void)hipFree(c->dp.compactMask);
    if (c->dp.bucketBits) (void)hipFree(c->dp.bucketBits);
    if (c->dp.allocMask) (void)hipFree(c->dp.allocMask);
    if (c->dp.macroBits) (void)hipFree(c->dp.macroBits);
    if (c->viewLists) (void)hipFree(c->viewLists);
    if (c->blockList) (void)hipFree(c->blockList);
    c->blockList = nullptr;
    c->viewLists = nullptr;
    c->viewListsSize = 0;
    c->dp = DevPtrs{};
    return VH_OK;
}

static int create_impl(const vh_config *cfg, uint32_t lo, uint32_t hi, vh_context **out)
{
    if (!cfg || !out) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    const HashTableParams &p = cfg->params;
    if (p.voxelBlockSize != 8) return fail(VH_ERR_INVALID_ARGUMENT, "voxelBlockSize must be 8");
    if (p.numBuckets == 0 || p.bucketSize == 0 || p.numVoxelBlocks == 0 || !(p.voxelSize > 0.0f))
        return fail(VH_ERR_INVALID_ARGUMENT, "numBuckets, bucketSize, numVoxelBlocks and voxelSize must be positive");
    if (cfg->width <= 0 || cfg->height <= 0 ||
        (uint64_t)((cfg->width + 15) / 16) * ((cfg->height + 15) / 16) * 256 > (1u << 21))
        return fail(VH_ERR_INVALID_ARGUMENT, "bad image size (at most 2^21 pixels in 16x16 tiles)");
    if (cfg->semantics != VH_SEM_REFERENCE && cfg->semantics != VH_SEM_PINHOLE)
        return fail(VH_ERR_INVALID_ARGUMENT, "bad semantics");
    if (lo >= hi || hi > p.numBuckets) return fail(VH_ERR_INVALID_ARGUMENT, "bad bucket range");
    if ((uint64_t)p.numVoxelBlocks * kBlockVoxels > 0x7fffffffull)
        return fail(VH_ERR_INVALID_ARGUMENT, "numVoxelBlocks*512 must fit the int ptr field");

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(VH_ERR_NO_DEVICE, "hipGetDeviceCount");
    int dev = cfg->device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return fail(VH_ERR_NO_DEVICE, "hipGetDevice");
    if (dev >= ndev) return fail(VH_ERR_INVALID_ARGUMENT, "device ordinal out of range");

    vh_context *c = new vh_context();
    c->device = dev;
    DeviceGuard guard(dev);
    if (!guard.ok) { delete c; return fail(VH_ERR_NO_DEVICE, "hipSetDevice"); }
    c->params = p;
    c->params.numOccupiedBlocks = 0;
    FrameParams &fp = c->fp;
    std::memcpy(fp.T, p.global_transform, sizeof fp.T);
    std::memcpy(fp.Tinv, p.inv_global_transform, sizeof fp.Tinv);
    fp.voxelSize = p.voxelSize;
    fp.truncation = p.truncation;
    fp.weightMax = p.integrationWeightMax;
    fp.width = cfg->width;
    fp.height = cfg->height;
    fp.semantics = cfg->semantics;
    fp.numBuckets = p.numBuckets;
    fp.bucketSize = p.bucketSize;
    fp.bucketLo = lo;
    fp.bucketHi = hi;
    fp.numVoxelBlocks = p.numVoxelBlocks;
    fp.epoch = 0;
    fp.allocBand = 0.0f;
    default_projection(c);

    c->ownedBuckets = hi - lo;
    c->numEntries = (size_t)c->ownedBuckets * p.bucketSize;
    const size_t npix = (size_t)cfg->width * cfg->height;
    DevPtrs &dp = c->dp;
    dp = DevPtrs{};
    dp.candCapacity = (uint32_t)npix;

#define VH_ALLOC(ptr, bytes)                                               \
    do {                                                                   \
        hipError_t e_ = hipMalloc((void **)&(ptr), (bytes));               \
        if (e_ != hipSuccess) {                                            \
            free_buffers(c);                                               \
            delete c;                                                      \
            return fail(e_ == hipErrorOutOfMemory ? VH_ERR_OUT_OF_MEMORY : VH_ERR_HIP, "hipMalloc " #ptr, e_); \
        }                                                                  \
    } while (0)
    VH_ALLOC(dp.heap, sizeof(uint32_t) * (size_t)p.numVoxelBlocks);
    VH_ALLOC(dp.table, sizeof(VoxelEntry) * c->numEntries + 16);   // the wide walk reads whole 16-byte chunks
    VH_ALLOC(dp.compact, sizeof(VoxelEntry) * c->numEntries);
    VH_ALLOC(dp.claim, sizeof(unsigned long long) * (size_t)c->ownedBuckets);
    VH_ALLOC(dp.blocks, sizeof(Voxel) * (size_t)p.numVoxelBlocks * kBlockVoxels);
    VH_ALLOC(dp.counters, sizeof(int32_t) * kNumCounters);
    VH_ALLOC(dp.candidates, sizeof(int4) * npix);
    VH_ALLOC(dp.compactMask, sizeof(uint32_t) * c->numEntries);
    VH_ALLOC(dp.bucketBits, sizeof(uint32_t) * (((size_t)c->ownedBuckets + 31) / 32));
    VH_ALLOC(dp.macroBits, kMacroBits / 8);
    // 32 mask words per 2048-entry tile, padded to whole 256-word chunks
    VH_ALLOC(dp.allocMask, sizeof(unsigned long long) *
                               (((c->numEntries + kMaskChunkEntries - 1) / kMaskChunkEntries) * kMaskChunkWords));
#undef VH_ALLOC

    // deviceAllocate, VoxelUtils.cu:183-208 (+ the compact table and the zeroed
    // volume the reference leaves to OpenGL)
    hipStream_t s = nullptr;
    const int g = 2048;
    reset_table_kernel<<<g, 256, 0, s>>>(dp.table, c->numEntries);
    reset_table_kernel<<<g, 256, 0, s>>>(dp.compact, c->numEntries);
    reset_heap_kernel<<<g, 256, 0, s>>>(dp.heap, p.numVoxelBlocks);
    hipError_t e = hipMemsetAsync(dp.claim, 0, sizeof(unsigned long long) * (size_t)c->ownedBuckets, s);
    if (e == hipSuccess) e = hipMemsetAsync(dp.bucketBits, 0, sizeof(uint32_t) * (((size_t)c->ownedBuckets + 31) / 32), s);
    if (e == hipSuccess) e = hipMemsetAsync(dp.macroBits, 0, kMacroBits / 8, s);
    if (e == hipSuccess) e = hipMemsetAsync(dp.blocks, 0, sizeof(Voxel) * (size_t)p.numVoxelBlocks * kBlockVoxels, s);
    int32_t h_counters[kNumCounters] = {0};
    h_counters[kHeapCounter] = (int32_t)p.numVoxelBlocks - 1;               // :207
    if (e == hipSuccess) e = hipMemcpyAsync(dp.counters, h_counters, sizeof h_counters, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) {
        free_buffers(c);
        delete c;
        return fail(VH_ERR_HIP, "table initialisation", e);
    }
    *out = c;
    return VH_OK;
}

extern "C" int vh_create(const vh_config *cfg, vh_context **out)
{
    if (!cfg) return fail(VH_ERR_INVALID_ARGUMENT, "null config");
    return create_impl(cfg, 0, cfg->params.numBuckets, out);
}

extern "C" int vh_create_shard(const vh_config *cfg, uint32_t lo, uint32_t hi, vh_context **out)
{
    return create_impl(cfg, lo, hi, out);
}

static void drop_events(vh_context *c)
{
    for (auto &t : c->timed) { (void)hipEventDestroy(t.start); (void)hipEventDestroy(t.stop); }
    c->timed.clear();
}

extern "C" int vh_destroy(vh_context *c)
{
    if (!c) return VH_OK;
    DeviceGuard guard(c->device);
    // the whole device, not c->stream: the caller's stream object may already be gone when a
    // language binding destroys the context late (interpreter exit)
    (void)hipDeviceSynchronize();
    drop_events(c);
    free_buffers(c);
    delete c;
    return VH_OK;
}

extern "C" int vh_set_stream(vh_context *c, void *stream)
{
    if (!c) return fail(VH_ERR_INVALID_ARGUMENT, "null context");
    c->stream = (hipStream_t)stream;
    return VH_OK;
}

extern "C" int vh_set_projection(vh_context *c, const float m[9])
{
    if (!c || !m) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    std::memcpy(c->fp.proj, m, 9 * sizeof(float));
    return VH_OK;
}

extern "C" int vh_set_raycast_intrinsics(vh_context *c, float fx, float fy, float cx, float cy)
{
    if (!c) return fail(VH_ERR_INVALID_ARGUMENT, "null context");
    c->rc_fx = fx; c->rc_fy = fy; c->rc_cx = cx; c->rc_cy = cy;
    return VH_OK;
}

extern "C" int vh_set_alloc_band(vh_context *c, float band_metres)
{
    if (!c || !(band_metres >= 0.0f)) return fail(VH_ERR_INVALID_ARGUMENT, "bad argument");
    if (band_metres / (4.0f * c->fp.voxelSize) > (float)((kMaxBandSamples - 1) / 2))
        return fail(VH_ERR_INVALID_ARGUMENT, "band wider than 31 half-block steps");
    c->fp.allocBand = band_metres;
    return VH_OK;
}

extern "C" int vh_set_pose(vh_context *c, const float pose[16])
{
    if (!c || !pose) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    std::memcpy(c->fp.T, pose, sizeof c->fp.T);
    invert4x4(pose, c->fp.Tinv);                        // SDF_Hashtable.cpp:15
    std::memcpy(c->params.global_transform, c->fp.T, sizeof c->fp.T);
    std::memcpy(c->params.inv_global_transform, c->fp.Tinv, sizeof c->fp.Tinv);
    return VH_OK;
}

// ---------------------------------------------------------------------------
// per-frame steps
// ---------------------------------------------------------------------------
extern "C" int vh_reset_mutexes(vh_context *c)
{
    if (!c) return fail(VH_ERR_INVALID_ARGUMENT, "null context");
    // The reference memsets 4*numBuckets bytes every frame (VoxelUtils.cu:146-149).
    // Claim words carry the epoch in their upper half, so starting a new epoch
    // invalidates every lock at once.  After 2^32-1 frames the words are cleared
    // for real and the epoch restarts.
    if (c->fp.epoch == 0xffffffffu) {
        DeviceGuard guard(c->device);
        VH_HIP(hipMemsetAsync(c->dp.claim, 0, sizeof(unsigned long long) * (size_t)c->ownedBuckets, c->stream));
        c->fp.epoch = 0;
    }
    c->fp.epoch += 1;
    return VH_OK;
}

static inline int grid_for(size_t n, int per_block) { return (int)((n + per_block - 1) / per_block); }

template <typename K, typename... Args>
static int launch(vh_context *c, int phase, K kernel, dim3 grid, dim3 block, Args... args)
{
    if (!c->profiling) {
        hipLaunchKernelGGL(kernel, grid, block, 0, c->stream, args...);
        return VH_OK;
    }
    TimedLaunch t{phase, nullptr, nullptr};
    VH_HIP(hipEventCreate(&t.start));
    VH_HIP(hipEventCreate(&t.stop));
    hipExtLaunchKernelGGL(kernel, grid, block, 0, c->stream, t.start, t.stop, 0, args...);
    c->timed.push_back(t);
    return VH_OK;
}

static int launch_alloc(vh_context *c, const vh_float4 *verts)
{
    const int npix = c->fp.width * c->fp.height;
    int rc = launch(c, kPhaseClaim, alloc_claim_kernel, dim3(grid_for(npix, 256)), dim3(256), c->fp, c->dp,
                    reinterpret_cast<const float4 *>(verts));
    if (rc != VH_OK) return rc;
    rc = launch(c, kPhaseCommit, alloc_commit_kernel, dim3(32), dim3(256), c->fp, c->dp);
    c->compactArmed = (rc == VH_OK);
    return rc;
}

// workgroups of the table walk: 2048 entries each (strided) or 2048 16-byte chunks each (wide)
static uint32_t walk_blocks(const vh_context *c)
{
    if (c->flattenVariant == kWalkWide)
        return (uint32_t)grid_for(((size_t)c->numEntries * 20 + 15) / 16, kFlattenThreads * kChunksPerLane);
    if (c->flattenVariant == kWalkIndexed)       // one lane per 32-bucket word of the occupancy bitmap
        return (uint32_t)grid_for(((size_t)c->ownedBuckets + 31) / 32, kFlattenThreads);
    if (c->flattenVariant == kWalkPersistent)    // resident workgroups striding over the tiles
        return std::min<uint32_t>((uint32_t)grid_for(c->numEntries, kFlattenThreads * kEntriesPerLane),
                                  (uint32_t)c->persistentBlocks);
    return (uint32_t)grid_for(c->numEntries, kFlattenThreads * kEntriesPerLane);
}

static int launch_flatten(vh_context *c)
{
    const dim3 grid(walk_blocks(c));
    if (c->flattenVariant == kWalkIndexed)
        return launch(c, kPhaseFlatten, flatten_kernel<kWalkIndexed>, grid, dim3(kFlattenThreads), c->fp, c->dp,
                      (uint32_t)c->numEntries);
    if (c->flattenVariant == kWalkPersistent)
        return launch(c, kPhaseFlatten, flatten_kernel<kWalkPersistent>, grid, dim3(kFlattenThreads), c->fp, c->dp,
                      (uint32_t)c->numEntries);
    if (c->flattenVariant == kWalkWide)
        return launch(c, kPhaseFlatten, flatten_kernel<kWalkWide>, grid, dim3(kFlattenThreads), c->fp, c->dp,
                      (uint32_t)c->numEntries);
    if (c->flattenVariant == kWalkStrided)
        return launch(c, kPhaseFlatten, flatten_kernel<kWalkStrided>, grid, dim3(kFlattenThreads), c->fp, c->dp,
                      (uint32_t)c->numEntries);
    if (c->flattenVariant == kWalkStridedNT)
        return launch(c, kPhaseFlatten, flatten_kernel<kWalkStridedNT>, grid, dim3(kFlattenThreads), c->fp, c->dp,
                      (uint32_t)c->numEntries);
    return launch(c, kPhaseFlatten, flatten_kernel<kWalkStridedBallot>, grid, dim3(kFlattenThreads), c->fp, c->dp,
                  (uint32_t)c->numEntries);
}

static int launch_integrate(vh_context *c, const vh_float4 *verts)
{
    return launch(c, kPhaseIntegrate, integrate_kernel, dim3(c->integrateGrid), dim3(256), c->fp, c->dp,
                  reinterpret_cast<const float4 *>(verts));
}

extern "C" int vh_alloc_blocks(vh_context *c, const vh_float4 *verts, const vh_float4 *normals)
{
    (void)normals;   // loaded into a dead variable by the reference (VoxelUtils.cu:631)
    if (!c || !verts) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    if (c->fp.epoch == 0) return fail(VH_ERR_INVALID_ARGUMENT, "vh_reset_mutexes must start the frame");
    DeviceGuard guard(c->device);
    int rc = launch_alloc(c, verts);
    if (rc != VH_OK) return rc;
    VH_HIP(hipGetLastError());
    return VH_OK;
}

extern "C" int vh_flatten(vh_context *c, int32_t *occupied_out)
{
    if (!c) return fail(VH_ERR_INVALID_ARGUMENT, "null context");
    DeviceGuard guard(c->device);
    if (!c->compactArmed)
        VH_HIP(hipMemsetAsync(c->dp.counters + kCompactCount, 0, sizeof(int32_t), c->stream));   // :760
    c->compactArmed = false;
    c->occupiedCounter = kCompactCount;
    int rc = launch_flatten(c);
    if (rc != VH_OK) return rc;
    VH_HIP(hipGetLastError());
    if (occupied_out) {
        int32_t n = 0;
        VH_HIP(hipMemcpyAsync(&n, c->dp.counters + kCompactCount, sizeof n, hipMemcpyDeviceToHost, c->stream));
        VH_HIP(hipStreamSynchronize(c->stream));                                               // :765
        *occupied_out = n;
        c->params.numOccupiedBlocks = (uint32_t)n;
    }
    return VH_OK;
}

extern "C" int vh_integrate_depth_map(vh_context *c, const vh_float4 *verts)
{
    if (!c || !verts) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    DeviceGuard guard(c->device);
    int rc = launch_integrate(c, verts);
    if (rc != VH_OK) return rc;
    VH_HIP(hipGetLastError());
    return VH_OK;
}

extern "C" int vh_integrate(vh_context *c, const float pose[16], const vh_float4 *verts, const vh_float4 *normals)
{
    (void)normals;
    if (!c || !pose || !verts) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    DeviceGuard guard(c->device);
    int rc = vh_set_pose(c, pose);
    if (rc == VH_OK) rc = vh_reset_mutexes(c);
    if (rc != VH_OK) return rc;
    if (c->fusedFrame && c->flattenVariant == kWalkMask) {
        // mask form: {claim || pure-stream walk that stores allocation masks}, then
        // {commit || consume the masks: frustum test, compaction, TSDF update}
        const uint32_t claimBlocks = (uint32_t)grid_for((size_t)c->fp.width * c->fp.height, 256);
        const uint32_t tiles = (uint32_t)grid_for(c->numEntries, kFlattenThreads * kEntriesPerLane);
        rc = launch(c, kPhaseFrameScanClaim, frame_mask_claim_kernel, dim3(claimBlocks + tiles), dim3(256), c->fp,
                    c->dp, reinterpret_cast<const float4 *>(verts), (uint32_t)c->numEntries, claimBlocks,
                    c->fusedParity);
        if (rc != VH_OK) return rc;
        const uint32_t commitBlocks = (uint32_t)c->commitBlocks;
        const uint32_t chunks = (uint32_t)grid_for(c->numEntries, kMaskChunkEntries);
        rc = launch(c, kPhaseFrameCommitIntegrate, frame_commit_consume_kernel, dim3(commitBlocks + chunks), dim3(256),
                    c->fp, c->dp, reinterpret_cast<const float4 *>(verts), (uint32_t)c->numEntries, commitBlocks,
                    c->fusedParity);
        if (rc != VH_OK) return rc;
        c->occupiedCounter = kScanCount + c->fusedParity;   // this frame's slot counter = occupied count
        c->fusedParity ^= 1;
        c->compactArmed = false;
    } else if (c->fusedFrame) {
        // two launches: {claim || table walk}, then {commit + integrate}; see vh_kernels.hip
        c->occupiedCounter = kCompactCount;
        const uint32_t claimBlocks = (uint32_t)grid_for((size_t)c->fp.width * c->fp.height, 256);
        const uint32_t scanBlocks = walk_blocks(c);
        if (c->flattenVariant == kWalkIndexed)
            rc = launch(c, kPhaseFrameScanClaim, frame_scan_claim_kernel<kWalkIndexed>,
                        dim3(claimBlocks + scanBlocks), dim3(256), c->fp, c->dp,
                        reinterpret_cast<const float4 *>(verts), (uint32_t)c->numEntries, claimBlocks,
                        c->fusedParity);
        else if (c->flattenVariant == kWalkPersistent)
            rc = launch(c, kPhaseFrameScanClaim, frame_scan_claim_kernel<kWalkPersistent>,
                        dim3(claimBlocks + scanBlocks), dim3(256), c->fp, c->dp,
                        reinterpret_cast<const float4 *>(verts), (uint32_t)c->numEntries, claimBlocks,
                        c->fusedParity);
        else if (c->flattenVariant == kWalkWide)
            rc = launch(c, kPhaseFrameScanClaim, frame_scan_claim_kernel<kWalkWide>, dim3(claimBlocks + scanBlocks),
                        dim3(256), c->fp, c->dp, reinterpret_cast<const float4 *>(verts), (uint32_t)c->numEntries,
                        claimBlocks, c->fusedParity);
        else if (c->flattenVariant == kWalkStridedBallot)
            rc = launch(c, kPhaseFrameScanClaim, frame_scan_claim_kernel<kWalkStridedBallot>,
                        dim3(claimBlocks + scanBlocks), dim3(256), c->fp, c->dp,
                        reinterpret_cast<const float4 *>(verts), (uint32_t)c->numEntries, claimBlocks,
                        c->fusedParity);
        else
            rc = launch(c, kPhaseFrameScanClaim, frame_scan_claim_kernel<kWalkStrided>,
                        dim3(claimBlocks + scanBlocks), dim3(256), c->fp, c->dp,
                        reinterpret_cast<const float4 *>(verts), (uint32_t)c->numEntries, claimBlocks,
                        c->fusedParity);
        if (rc != VH_OK) return rc;
        const uint32_t commitBlocks = (uint32_t)c->commitBlocks;
        rc = launch(c, kPhaseFrameCommitIntegrate, frame_commit_integrate_kernel,
                    dim3(commitBlocks + (uint32_t)c->integrateGrid), dim3(256), c->fp, c->dp,
                    reinterpret_cast<const float4 *>(verts), commitBlocks, c->fusedParity);
        if (rc != VH_OK) return rc;
        c->fusedParity ^= 1;       // this frame cleared the other counter set for the next one
        c->compactArmed = false;
    } else {
        // alloc_commit re-arms the compact counter, so no memset node is needed here
        c->occupiedCounter = kCompactCount;
        if ((rc = launch_alloc(c, verts)) != VH_OK) return rc;
        c->compactArmed = false;
        if ((rc = launch_flatten(c)) != VH_OK) return rc;
        if ((rc = launch_integrate(c, verts)) != VH_OK) return rc;
    }
    if (c->profiling) c->profiledFrames += 1;
    VH_HIP(hipGetLastError());
    return VH_OK;
}

// The frame straight from the uint16 sensor image: preProcess's vertex computation happens inside
// the claim half, the TSDF update reads the image.  Equals vh_preprocess + vh_integrate.
extern "C" int vh_integrate_depth(vh_context *c, const float pose[16], const uint16_t *d_depth, const float k_inv[9])
{
    if (!c || !pose || !d_depth || !k_inv) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    DeviceGuard guard(c->device);
    int rc = vh_set_pose(c, pose);
    if (rc == VH_OK) rc = vh_reset_mutexes(c);
    if (rc != VH_OK) return rc;
    SensorImage in;
    in.depth = d_depth;
    std::memcpy(in.k, k_inv, sizeof in.k);
    in.unit = 5000.0f;                                                   // CameraTrackingUtils.cu:64
    c->occupiedCounter = kCompactCount;
    const uint32_t claimBlocks = (uint32_t)grid_for((size_t)c->fp.width * c->fp.height, 256);
    const uint32_t scanBlocks = (uint32_t)grid_for(c->numEntries, kFlattenThreads * kEntriesPerLane);
    rc = launch(c, kPhaseFrameScanClaim, frame_scan_claim_sensor_kernel, dim3(claimBlocks + scanBlocks), dim3(256), c->fp,
                c->dp, in, (uint32_t)c->numEntries, claimBlocks, c->fusedParity);
    if (rc != VH_OK) return rc;
    const uint32_t commitBlocks = (uint32_t)c->commitBlocks;
    rc = launch(c, kPhaseFrameCommitIntegrate, frame_commit_integrate_sensor_kernel,
                dim3(commitBlocks + (uint32_t)c->integrateGrid), dim3(256), c->fp, c->dp, in, commitBlocks, c->fusedParity);
    if (rc != VH_OK) return rc;
    c->fusedParity ^= 1;
    c->compactArmed = false;
    if (c->profiling) c->profiledFrames += 1;
    VH_HIP(hipGetLastError());
    return VH_OK;
}

extern "C" int vh_raycast(vh_context *c, const float pose[16], float t_min, float t_max, float *d_depth_out)
{
    if (!c || !pose || !d_depth_out) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    if (!(t_max > t_min)) return fail(VH_ERR_INVALID_ARGUMENT, "t_max must exceed t_min");
    DeviceGuard guard(c->device);
    FrameParams fp = c->fp;
    std::memcpy(fp.T, pose, sizeof fp.T);
    const float q = (t_max - t_min) / fp.voxelSize;
    int nsteps = (q >= 2147483648.0f) ? 0x7fffffff : (int)q;
    nsteps += 1;
    dim3 grid((fp.width + 15) / 16, (fp.height + 15) / 16);
    DevPtrs dp = c->dp;
    if (c->viewBlocks) dp.blocks = const_cast<Voxel *>(c->viewBlocks);     // view table: voxels live in the records
    const int rc = c->raycastPatch
                       ? launch(c, kPhaseRaycast, raycast_kernel<1>, grid, dim3(256), fp, dp, c->rc_fx, c->rc_fy,
                                c->rc_cx, c->rc_cy, t_min, nsteps, d_depth_out, c->raycastXcd)
                       : launch(c, kPhaseRaycast, raycast_kernel<0>, grid, dim3(256), fp, dp, c->rc_fx, c->rc_fy,
                                c->rc_cx, c->rc_cy, t_min, nsteps, d_depth_out, c->raycastXcd);
    if (rc != VH_OK) return rc;
    VH_HIP(hipGetLastError());
    return VH_OK;
}

// Block silhouettes: SDFRenderer::drawToFrontAndBack, SDFRenderer.cpp:165-208.
extern "C" int vh_render_blocks(vh_context *c, const float pose[16], float t_min, float t_max, float *d_front,
                                float *d_back)
{
    if (!c || !pose || !d_front || !d_back) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    if (!(t_max > t_min) || !(t_min >= 0.0f)) return fail(VH_ERR_INVALID_ARGUMENT, "need 0 <= t_min < t_max");
    DeviceGuard guard(c->device);
    BlockView bv;
    float inv[16];
    invert4x4(pose, inv);
    std::memcpy(bv.T, pose, sizeof bv.T);
    std::memcpy(bv.Tinv, inv, sizeof bv.Tinv);
    bv.fx = c->rc_fx; bv.fy = c->rc_fy; bv.cx = c->rc_cx; bv.cy = c->rc_cy;
    bv.tMin = t_min;
    bv.tMax = t_max;
    const int32_t npix = c->fp.width * c->fp.height;
    // list of the allocated entries: room for every entry of the table, allocated on first use (synchronises once)
    if (!c->blockList) {
        VH_HIP(hipStreamSynchronize(c->stream));
        VH_HIP(hipMalloc((void **)&c->blockList, sizeof(int32_t) * (c->numEntries + 4)));
    }
    int32_t *listCount = c->blockList;
    int32_t *list = listCount + 4;
    const int32_t capacity = (int32_t)c->numEntries;
    uint32_t *front = reinterpret_cast<uint32_t *>(d_front), *back = reinterpret_cast<uint32_t *>(d_back);
    int rc = launch(c, kPhaseRaycastBounds, blocks_init_kernel, dim3((unsigned)grid_for((size_t)npix, 256)), dim3(256), front,
                    back, npix, listCount);
    const uint32_t words = (c->ownedBuckets + 31u) / 32u;
    if (rc == VH_OK)
        rc = launch(c, kPhaseRaycastBounds, blocks_list_kernel, dim3((unsigned)grid_for(words, 256)), dim3(256), c->fp, c->dp,
                    list, capacity, listCount);
    if (rc == VH_OK)
        rc = launch(c, kPhaseRaycastBounds, blocks_raster_kernel, dim3(1024, 16), dim3(256), c->fp, c->dp, bv,
                    (const int32_t *)list, capacity, (const int32_t *)listCount, front, back);
    if (rc == VH_OK)
        rc = launch(c, kPhaseRaycastBounds, blocks_finish_kernel, dim3((unsigned)grid_for((size_t)npix, 256)), dim3(256), front,
                    npix);
    if (rc != VH_OK) return rc;
    VH_HIP(hipGetLastError());
    return VH_OK;
}

// ---------------------------------------------------------------------------
// block deletion / garbage collection
// ---------------------------------------------------------------------------
static int sweep_and_release(vh_context *c)
{
    int rc = launch(c, kPhaseGc, gc_sweep_kernel, dim3(256), dim3(256), c->fp, c->dp);
    if (rc != VH_OK) return rc;
    rc = launch(c, kPhaseGc, gc_release_kernel, dim3(1024), dim3(256), c->dp);
    if (rc != VH_OK) return rc;
    rc = launch(c, kPhaseGc, gc_finish_kernel, dim3(1), dim3(1), c->dp, c->occupiedCounter);
    if (rc != VH_OK) return rc;
    if (c->profiling) c->times.gc_calls += 1;
    c->params.numOccupiedBlocks = 0;
    VH_HIP(hipGetLastError());
    return VH_OK;
}

extern "C" int vh_delete_blocks(vh_context *c, const int32_t *d_keys, int32_t n)
{
    if (!c || (!d_keys && n > 0) || n < 0) return fail(VH_ERR_INVALID_ARGUMENT, "bad argument");
    if (c->viewBlocks) return fail(VH_ERR_INVALID_ARGUMENT, "a view table owns no blocks");
    DeviceGuard guard(c->device);
    c->fp.epoch += 1;                       // the sweep list is built under a fresh lock epoch
    if (n > 0) {
        const int rc = launch(c, kPhaseGc, gc_mark_keys_kernel, dim3((unsigned)grid_for((size_t)n, 256)), dim3(256), c->fp,
                              c->dp, reinterpret_cast<const int4 *>(d_keys), n);
        if (rc != VH_OK) return rc;
    }
    return sweep_and_release(c);
}

extern "C" int vh_garbage_collect(vh_context *c, float sdf_threshold)
{
    if (!c) return fail(VH_ERR_INVALID_ARGUMENT, "null context");
    if (c->viewBlocks) return fail(VH_ERR_INVALID_ARGUMENT, "a view table owns no blocks");
    DeviceGuard guard(c->device);
    c->fp.epoch += 1;
    const int rc = launch(c, kPhaseGc, gc_identify_kernel, dim3(2048), dim3(256), c->fp, c->dp, c->occupiedCounter,
                          sdf_threshold);
    if (rc != VH_OK) return rc;
    return sweep_and_release(c);
}

// ---------------------------------------------------------------------------
// raycast over shards: export of the blocks a view can touch, import into a view table
// ---------------------------------------------------------------------------
// oracle: vho_view_frustum (same operations in the same order)
static void make_view_frustum(const vh_context *c, const float pose[16], float t_min, float t_max, float f[22])
{
    float inv[16];
    invert4x4(pose, inv);
    std::memcpy(f, inv, 12 * sizeof(float));
    const float r = 7.0f * c->fp.voxelSize;
    const float a[4] = {(0.0f - c->rc_cx) / c->rc_fx, ((float)(c->fp.width - 1) - c->rc_cx) / c->rc_fx,
                        (0.0f - c->rc_cy) / c->rc_fy, ((float)(c->fp.height - 1) - c->rc_cy) / c->rc_fy};
    for (int i = 0; i < 4; ++i) {
        f[12 + i] = a[i];
        f[16 + i] = -(r * sqrtf(1.0f + a[i] * a[i]));
    }
    f[20] = t_min - r;
    f[21] = t_max + r;
}

extern "C" int vh_export_views(vh_context *c, const float *poses, int32_t n_views, float t_min, float t_max,
                               vh_view_record *d_records, int32_t capacity, int32_t *d_counts)
{
    if (!c || !poses || !d_records || !d_counts) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    if (n_views < 1 || n_views > VH_MAX_CAMERAS) return fail(VH_ERR_INVALID_ARGUMENT, "1..VH_MAX_CAMERAS views");
    if (capacity < 1) return fail(VH_ERR_INVALID_ARGUMENT, "capacity must be positive");
    if (!(t_max > t_min)) return fail(VH_ERR_INVALID_ARGUMENT, "t_max must exceed t_min");
    if (c->viewBlocks) return fail(VH_ERR_INVALID_ARGUMENT, "a view table has no voxels of its own to export");
    DeviceGuard guard(c->device);
    const size_t need = (size_t)n_views * (size_t)capacity;
    if (c->viewListsSize < need) {                         // first call (or a larger one): synchronises
        VH_HIP(hipStreamSynchronize(c->stream));
        if (c->viewLists) (void)hipFree(c->viewLists);
        c->viewLists = nullptr;
        c->viewListsSize = 0;
        VH_HIP(hipMalloc((void **)&c->viewLists, need * sizeof(int32_t)));
        c->viewListsSize = need;
    }
    VH_HIP(hipMemsetAsync(d_counts, 0, sizeof(int32_t) * (size_t)n_views, c->stream));
    const uint32_t tiles = (uint32_t)((c->numEntries + kFlattenThreads * kEntriesPerLane - 1) /
                                      (kFlattenThreads * kEntriesPerLane));
    for (int32_t base = 0; base < n_views; base += kMaxViewsPerLaunch) {
        const int32_t n = std::min<int32_t>(kMaxViewsPerLaunch, n_views - base);
        ViewSet vs;
        std::memset(&vs, 0, sizeof vs);
        for (int32_t v = 0; v < n; ++v) make_view_frustum(c, poses + 16 * (size_t)(base + v), t_min, t_max, vs.v[v].f);
        const int rc = launch(c, kPhaseViewExport, view_select_kernel, dim3(tiles), dim3(kFlattenThreads), c->fp, c->dp,
                              (uint32_t)c->numEntries, vs, n, c->viewLists + (size_t)base * capacity, capacity,
                              d_counts + base);
        if (rc != VH_OK) return rc;
    }
    const int rc = launch(c, kPhaseViewExport, view_pack_kernel, dim3((unsigned)std::min<int32_t>(capacity, 2048), n_views),
                          dim3(256), c->dp, (const int32_t *)c->viewLists, (const int32_t *)d_counts, capacity,
                          reinterpret_cast<uint8_t *>(d_records));
    if (rc != VH_OK) return rc;
    VH_HIP(hipGetLastError());
    return VH_OK;
}

extern "C" int vh_import_view(vh_context *c, const vh_view_record *d_records, int32_t count)
{
    if (!c || (!d_records && count > 0)) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    if (count < 0 || (size_t)count > c->numEntries || (uint64_t)count * kViewRecordVoxels + 514ull > 0x7fffffffull)
        return fail(VH_ERR_INVALID_ARGUMENT, "bad record count");
    if (c->fp.bucketLo != 0 || c->fp.bucketHi != c->fp.numBuckets)
        return fail(VH_ERR_INVALID_ARGUMENT, "a view table is unsharded");
    if (c->fp.epoch != 0) return fail(VH_ERR_INVALID_ARGUMENT, "this context has integrated frames: use a dedicated view context");
    DeviceGuard guard(c->device);
    if (c->viewCount > 0) {
        const int rc = launch(c, kPhaseViewImport, view_clear_kernel, dim3((unsigned)grid_for((size_t)c->viewCount, 256)),
                              dim3(256), c->fp, c->dp, c->viewCount);
        if (rc != VH_OK) return rc;
    }
    VH_HIP(hipMemsetAsync(c->dp.bucketBits, 0, sizeof(uint32_t) * (((size_t)c->ownedBuckets + 31) / 32), c->stream));
    VH_HIP(hipMemsetAsync(c->dp.macroBits, 0, kMacroBits / 8, c->stream));
    c->viewBlocks = reinterpret_cast<const Voxel *>(d_records);
    c->viewCount = count;
    if (count > 0) {
        const int rc = launch(c, kPhaseViewImport, view_import_kernel, dim3((unsigned)grid_for((size_t)count, 256)),
                              dim3(256), c->fp, c->dp, reinterpret_cast<const uint8_t *>(d_records), count);
        if (rc != VH_OK) return rc;
    }
    VH_HIP(hipGetLastError());
    return VH_OK;
}

// ---------------------------------------------------------------------------
// sharding
// ---------------------------------------------------------------------------
// 4-byte units of one camera packet in the context's packet format
static size_t packet_units(const vh_context *c)
{
    const size_t npix = (size_t)c->fp.width * c->fp.height;
    return c->packetFormat == VH_PACKET_U16 ? (size_t)kPacketHeaderU16 + (npix + 1) / 2 : (size_t)kPacketHeader + npix;
}

extern "C" int vh_generate_keys(vh_context *c, const vh_float4 *verts, uint32_t camera_id, int32_t num_shards,
                                int32_t *d_bins, int32_t capacity, int32_t bin_stride, float *d_packet)
{
    if (bin_stride == 0) bin_stride = capacity;
    if (!c || !verts || !d_bins || num_shards <= 0 || capacity < 2 || bin_stride < capacity ||
        camera_id >= VH_MAX_CAMERAS || num_shards > VH_MAX_CAMERAS)
        return fail(VH_ERR_INVALID_ARGUMENT, "bad argument");
    DeviceGuard guard(c->device);
    const int npix = c->fp.width * c->fp.height;
    prepare_bins_kernel<<<1, 64, 0, c->stream>>>(reinterpret_cast<int4 *>(d_bins), num_shards, bin_stride, 1, 0);
    generate_keys_kernel<<<grid_for(npix, kGenThreads), kGenThreads, 0, c->stream>>>(
        c->fp, reinterpret_cast<const float4 *>(verts), num_shards, reinterpret_cast<int4 *>(d_bins), capacity,
        bin_stride, d_packet ? d_packet + kPacketHeader : nullptr, camera_id << kRankCameraShift);
    VH_HIP(hipGetLastError());
    return VH_OK;
}

// `batch` frames of one camera in one call: one launch zeroes all bin headers, then one launch per
// kGenBatch frames (blockIdx.y = frame, poses and vertex-map pointers in the kernel arguments).
extern "C" int vh_generate_keys_batch(vh_context *c, int32_t batch, const float *poses,
                                      const vh_float4 *const *d_verts, uint32_t camera_id, int32_t num_shards,
                                      int32_t *d_bins, int32_t capacity, int32_t bin_stride, int32_t frame_stride,
                                      float *d_packets, size_t packet_frame_stride)
{
    if (!c || !poses || !d_verts || !d_bins || batch <= 0 || num_shards <= 0 || num_shards > VH_MAX_CAMERAS ||
        capacity < 2 || camera_id >= VH_MAX_CAMERAS)
        return fail(VH_ERR_INVALID_ARGUMENT, "bad argument");
    if (frame_stride == 0) frame_stride = capacity;
    if (bin_stride == 0) bin_stride = batch * frame_stride;
    const size_t dense = (size_t)kPacketHeader + (size_t)c->fp.width * c->fp.height;
    if (packet_frame_stride == 0) packet_frame_stride = dense;
    if (frame_stride < capacity || bin_stride < batch * frame_stride || (d_packets && packet_frame_stride < dense))
        return fail(VH_ERR_INVALID_ARGUMENT, "bad stride");
    DeviceGuard guard(c->device);
    const int npix = c->fp.width * c->fp.height;
    prepare_bins_kernel<<<grid_for((size_t)num_shards * batch, 256), 256, 0, c->stream>>>(
        reinterpret_cast<int4 *>(d_bins), num_shards, bin_stride, batch, frame_stride);
    for (int b0 = 0; b0 < batch; b0 += kGenBatch) {
        const int n = std::min<int>(kGenBatch, batch - b0);
        GenFrames fr;
        std::memset(&fr, 0, sizeof fr);
        for (int j = 0; j < n; ++j) {
            const int rc = vh_set_pose(c, poses + 16 * (size_t)(b0 + j));        // pose + cofactor inverse
            if (rc != VH_OK) return rc;
            if (!d_verts[b0 + j]) return fail(VH_ERR_INVALID_ARGUMENT, "null vertex map");
            std::memcpy(fr.T[j], c->fp.T, sizeof fr.T[j]);
            std::memcpy(fr.Tinv[j], c->fp.Tinv, sizeof fr.Tinv[j]);
            fr.verts[j] = reinterpret_cast<const float4 *>(d_verts[b0 + j]);
        }
        generate_keys_batch_kernel<<<dim3((unsigned)grid_for(npix, kGenThreads), (unsigned)n), kGenThreads, 0, c->stream>>>(
            c->fp, fr, num_shards, reinterpret_cast<int4 *>(d_bins) + (size_t)frame_stride * b0, capacity, bin_stride,
            frame_stride, d_packets ? d_packets + packet_frame_stride * (size_t)b0 : nullptr, packet_frame_stride,
            camera_id << kRankCameraShift);
    }
    VH_HIP(hipGetLastError());
    return VH_OK;
}

// Keys and sensor-depth packets of `batch` frames of this camera from the uint16 images alone: one
// launch per kGenBatch frames (no vertex map, no separate packet pass).
extern "C" int vh_generate_keys_depth_batch(vh_context *c, int32_t batch, const float *poses,
                                            const uint16_t *const *d_depth, const float k_inv[9], uint32_t camera_id,
                                            int32_t num_shards, int32_t *d_bins, int32_t capacity, int32_t bin_stride,
                                            int32_t frame_stride, float *d_packets, size_t packet_frame_stride)
{
    if (!c || !poses || !d_depth || !k_inv || !d_bins || batch <= 0 || num_shards <= 0 ||
        num_shards > VH_MAX_CAMERAS || capacity < 2 || camera_id >= VH_MAX_CAMERAS)
        return fail(VH_ERR_INVALID_ARGUMENT, "bad argument");
    const size_t npix = (size_t)c->fp.width * c->fp.height;
    if (d_packets && npix % 2) return fail(VH_ERR_INVALID_ARGUMENT, "sensor-depth packets need an even number of pixels");
    if (frame_stride == 0) frame_stride = capacity;
    if (bin_stride == 0) bin_stride = batch * frame_stride;
    const size_t dense = (size_t)kPacketHeaderU16 + npix / 2;
    if (packet_frame_stride == 0) packet_frame_stride = dense;
    if (frame_stride < capacity || bin_stride < batch * frame_stride || (d_packets && packet_frame_stride < dense))
        return fail(VH_ERR_INVALID_ARGUMENT, "bad stride");
    DeviceGuard guard(c->device);
    prepare_bins_kernel<<<grid_for((size_t)num_shards * batch, 256), 256, 0, c->stream>>>(
        reinterpret_cast<int4 *>(d_bins), num_shards, bin_stride, batch, frame_stride);
    for (int b0 = 0; b0 < batch; b0 += kGenBatch) {
        const int n = std::min<int>(kGenBatch, batch - b0);
        GenSensorFrames fr;
        std::memset(&fr, 0, sizeof fr);
        std::memcpy(fr.k, k_inv, sizeof fr.k);
        fr.unit = 5000.0f;                                               // CameraTrackingUtils.cu:64
        for (int j = 0; j < n; ++j) {
            const int rc = vh_set_pose(c, poses + 16 * (size_t)(b0 + j));
            if (rc != VH_OK) return rc;
            if (!d_depth[b0 + j]) return fail(VH_ERR_INVALID_ARGUMENT, "null depth image");
            std::memcpy(fr.T[j], c->fp.T, sizeof fr.T[j]);
            std::memcpy(fr.Tinv[j], c->fp.Tinv, sizeof fr.Tinv[j]);
            fr.depth[j] = d_depth[b0 + j];
        }
        generate_keys_sensor_batch_kernel<<<dim3((unsigned)grid_for(npix, kGenThreads), (unsigned)n), kGenThreads, 0,
                                            c->stream>>>(
            c->fp, fr, num_shards, reinterpret_cast<int4 *>(d_bins) + (size_t)frame_stride * b0, capacity, bin_stride,
            frame_stride, d_packets ? d_packets + packet_frame_stride * (size_t)b0 : nullptr, packet_frame_stride,
            camera_id << kRankCameraShift);
    }
    VH_HIP(hipGetLastError());
    return VH_OK;
}

// Sensor-depth packets (VH_PACKET_U16) of `batch` frames of this camera: pose + inverse + K_inv row 2 +
// depth unit, then the uint16 image as it is.  The key generation for the same frames is
// vh_generate_keys_batch with d_packets = NULL.
extern "C" int vh_write_packets_u16_batch(vh_context *c, int32_t batch, const float *poses,
                                          const uint16_t *const *d_depth, const float k_inv[9], float *d_packets,
                                          size_t packet_frame_stride)
{
    if (!c || !poses || !d_depth || !k_inv || !d_packets || batch <= 0)
        return fail(VH_ERR_INVALID_ARGUMENT, "bad argument");
    const size_t npix = (size_t)c->fp.width * c->fp.height;
    if (npix % 2) return fail(VH_ERR_INVALID_ARGUMENT, "sensor-depth packets need an even number of pixels");
    const size_t dense = (size_t)kPacketHeaderU16 + npix / 2;
    if (packet_frame_stride == 0) packet_frame_stride = dense;
    if (packet_frame_stride < dense) return fail(VH_ERR_INVALID_ARGUMENT, "bad stride");
    DeviceGuard guard(c->device);
    for (int b0 = 0; b0 < batch; b0 += kGenBatch) {
        const int n = std::min<int>(kGenBatch, batch - b0);
        SensorFrames fr;
        std::memset(&fr, 0, sizeof fr);
        fr.k6 = k_inv[6]; fr.k7 = k_inv[7]; fr.k8 = k_inv[8];
        fr.unit = 5000.0f;                                               // CameraTrackingUtils.cu:64
        for (int j = 0; j < n; ++j) {
            const int rc = vh_set_pose(c, poses + 16 * (size_t)(b0 + j));
            if (rc != VH_OK) return rc;
            if (!d_depth[b0 + j]) return fail(VH_ERR_INVALID_ARGUMENT, "null depth image");
            std::memcpy(fr.T[j], c->fp.T, sizeof fr.T[j]);
            std::memcpy(fr.Tinv[j], c->fp.Tinv, sizeof fr.Tinv[j]);
            fr.depth[j] = d_depth[b0 + j];
        }
        write_packets_u16_kernel<<<dim3(64, (unsigned)n), 256, 0, c->stream>>>(
            fr, (int32_t)npix, d_packets + packet_frame_stride * (size_t)b0, packet_frame_stride);
    }
    VH_HIP(hipGetLastError());
    return VH_OK;
}

// `batch` multi-camera frames applied one after the other, each as the fused pair of launches
// (new lock epoch; {claim bins || walk}; {commit + integrate}).
extern "C" int vh_apply_frames_batch(vh_context *c, int32_t batch, const int32_t *d_bins, int32_t num_bins,
                                     int32_t capacity, int32_t bin_stride, int32_t frame_stride, int32_t num_cams,
                                     const float *d_packets, size_t packet_stride, size_t packet_frame_stride)
{
    if (!c || !d_bins || !d_packets || batch <= 0 || num_bins <= 0 || capacity < 2 || num_cams <= 0 ||
        num_cams > VH_MAX_CAMERAS)
        return fail(VH_ERR_INVALID_ARGUMENT, "bad argument");
    const size_t dense = packet_units(c);
    if (frame_stride == 0) frame_stride = capacity;
    if (bin_stride == 0) bin_stride = batch * frame_stride;
    if (packet_frame_stride == 0) packet_frame_stride = dense;
    if (packet_stride == 0) packet_stride = (size_t)batch * packet_frame_stride;
    if (frame_stride < capacity || bin_stride < batch * frame_stride || packet_frame_stride < dense ||
        packet_stride < (size_t)batch * packet_frame_stride)
        return fail(VH_ERR_INVALID_ARGUMENT, "bad stride");
    DeviceGuard guard(c->device);
    uint32_t parts = (uint32_t)grid_for((size_t)capacity, 256 * 4);
    if (parts < 1) parts = 1;
    const uint32_t scanBlocks = (uint32_t)grid_for(c->numEntries, kFlattenThreads * kEntriesPerLane);
    const uint32_t commitBlocks = (uint32_t)c->commitBlocks;
    for (int b = 0; b < batch; ++b) {
        int rc = vh_reset_mutexes(c);
        if (rc != VH_OK) return rc;
        const int4 *bins = reinterpret_cast<const int4 *>(d_bins) + (size_t)frame_stride * b;
        const float *packets = d_packets + packet_frame_stride * b;
        rc = launch(c, kPhaseFrameScanClaim, frame_multi_scan_claim_kernel,
                    dim3((uint32_t)num_bins * parts + scanBlocks), dim3(256), c->fp, c->dp, bins, capacity, bin_stride,
                    (uint32_t)num_bins, parts, (uint32_t)c->numEntries, num_cams, packets, packet_stride,
                    c->fusedParity);
        if (rc == VH_OK)
            rc = c->packetFormat == VH_PACKET_U16
                     ? launch(c, kPhaseFrameCommitIntegrate, frame_multi_commit_integrate_kernel<true>,
                              dim3(commitBlocks + (uint32_t)c->integrateGrid), dim3(256), c->fp, c->dp, num_cams,
                              packets, packet_stride, commitBlocks, c->fusedParity)
                     : launch(c, kPhaseFrameCommitIntegrate, frame_multi_commit_integrate_kernel<false>,
                              dim3(commitBlocks + (uint32_t)c->integrateGrid), dim3(256), c->fp, c->dp, num_cams,
                              packets, packet_stride, commitBlocks, c->fusedParity);
        if (rc != VH_OK) return rc;
        c->fusedParity ^= 1;
        c->compactArmed = false;
        c->occupiedCounter = kCompactCount;
        if (c->profiling) c->profiledFrames += 1;
    }
    VH_HIP(hipGetLastError());
    return VH_OK;
}

extern "C" int vh_insert_bins(vh_context *c, const int32_t *d_bins, int32_t num_bins, int32_t capacity,
                              int32_t bin_stride)
{
    if (bin_stride == 0) bin_stride = capacity;
    if (!c || !d_bins || num_bins <= 0 || capacity < 2 || bin_stride < capacity)
        return fail(VH_ERR_INVALID_ARGUMENT, "bad argument");
    if (c->fp.epoch == 0) return fail(VH_ERR_INVALID_ARGUMENT, "vh_reset_mutexes must start the frame");
    DeviceGuard guard(c->device);
    int gx = grid_for((size_t)capacity, 256 * 4);
    if (gx < 1) gx = 1;
    int rc = launch(c, kPhaseClaim, claim_bins_kernel, dim3(gx, num_bins), dim3(256), c->fp, c->dp,
                    reinterpret_cast<const int4 *>(d_bins), capacity, bin_stride);
    if (rc == VH_OK) rc = launch(c, kPhaseCommit, alloc_commit_kernel, dim3(32), dim3(256), c->fp, c->dp);
    if (rc != VH_OK) return rc;
    c->compactArmed = true;
    VH_HIP(hipGetLastError());
    return VH_OK;
}

extern "C" int vh_integrate_packets(vh_context *c, int32_t num_cams, const float *d_packets, size_t packet_stride)
{
    const size_t dense = c ? packet_units(c) : 0;
    if (packet_stride == 0) packet_stride = dense;
    if (!c || !d_packets || num_cams <= 0 || num_cams > VH_MAX_CAMERAS || packet_stride < dense)
        return fail(VH_ERR_INVALID_ARGUMENT, "bad argument");
    DeviceGuard guard(c->device);
    const size_t stride = packet_stride;
    if (!c->compactArmed)
        VH_HIP(hipMemsetAsync(c->dp.counters + kCompactCount, 0, sizeof(int32_t), c->stream));
    c->compactArmed = false;
    c->occupiedCounter = kCompactCount;
    int rc = launch(c, kPhaseFlatten, flatten_multi_kernel,
                    dim3(grid_for(c->numEntries, kFlattenThreads * kEntriesPerLane)), dim3(kFlattenThreads), c->fp,
                    c->dp, (uint32_t)c->numEntries, num_cams, d_packets, stride);
    if (rc == VH_OK)
        rc = c->packetFormat == VH_PACKET_U16
                 ? launch(c, kPhaseIntegrate, integrate_multi_kernel<true>, dim3(c->integrateGrid), dim3(256), c->fp, c->dp,
                          num_cams, d_packets, stride)
                 : launch(c, kPhaseIntegrate, integrate_multi_kernel<false>, dim3(c->integrateGrid), dim3(256), c->fp,
                          c->dp, num_cams, d_packets, stride);
    if (rc != VH_OK) return rc;
    if (c->profiling) c->profiledFrames += 1;
    VH_HIP(hipGetLastError());
    return VH_OK;
}

// ---------------------------------------------------------------------------
// queries
// ---------------------------------------------------------------------------
extern "C" int vh_synchronize(vh_context *c)
{
    if (!c) return fail(VH_ERR_INVALID_ARGUMENT, "null context");
    DeviceGuard guard(c->device);
    VH_HIP(hipStreamSynchronize(c->stream));
    return VH_OK;
}

extern "C" int vh_get_counters(vh_context *c, vh_counters *out)
{
    if (!c || !out) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    DeviceGuard guard(c->device);
    int32_t h[kNumCounters];
    VH_HIP(hipMemcpyAsync(h, c->dp.counters, sizeof h, hipMemcpyDeviceToHost, c->stream));
    VH_HIP(hipStreamSynchronize(c->stream));
    out->occupied = h[c->occupiedCounter];
    out->heap_counter = h[kHeapCounter];
    out->allocated_total = (uint32_t)h[kAllocatedTotal];
    out->heap_exhausted = (uint32_t)h[kHeapExhausted];
    out->candidates = (uint32_t)h[kLastCandidates];
    out->epoch = c->fp.epoch;
    out->bin_overflow = (uint32_t)h[kBinOverflow];
    out->freed_total = (uint32_t)h[kFreedTotal];
    out->last_freed = (uint32_t)h[kLastFreed];
    c->params.numOccupiedBlocks = (uint32_t)h[c->occupiedCounter];
    return VH_OK;
}

extern "C" int vh_get_params(vh_context *c, HashTableParams *out)
{
    if (!c || !out) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    *out = c->params;
    return VH_OK;
}

extern "C" int vh_get_device_pointers(vh_context *c, PtrContainer *out)
{
    if (!c || !out) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    out->d_heap = c->dp.heap;
    out->d_hashTable = c->dp.table;
    out->d_compactifiedHashTable = c->dp.compact;
    out->d_hashTableBucketMutex = reinterpret_cast<uint64_t *>(c->dp.claim);
    out->d_SDFBlocks = c->dp.blocks;
    out->d_heapCounter = c->dp.counters + kHeapCounter;
    out->d_compactifiedHashCounter = c->dp.counters + kCompactCount;
    return VH_OK;
}

extern "C" int vh_download(vh_context *c, int which, void *dst, size_t bytes)
{
    if (!c || !dst) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    const void *src = nullptr;
    size_t avail = 0;
    switch (which) {
        case VH_BUF_HASH_TABLE: src = c->dp.table; avail = sizeof(VoxelEntry) * c->numEntries; break;
        case VH_BUF_COMPACT: src = c->dp.compact; avail = sizeof(VoxelEntry) * c->numEntries; break;
        case VH_BUF_SDF_BLOCKS: src = c->dp.blocks; avail = sizeof(Voxel) * (size_t)c->params.numVoxelBlocks * kBlockVoxels; break;
        case VH_BUF_HEAP: src = c->dp.heap; avail = sizeof(uint32_t) * (size_t)c->params.numVoxelBlocks; break;
        default: return fail(VH_ERR_INVALID_ARGUMENT, "unknown buffer id");
    }
    if (bytes > avail) return fail(VH_ERR_INVALID_ARGUMENT, "download larger than the buffer");
    DeviceGuard guard(c->device);
    VH_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    VH_HIP(hipStreamSynchronize(c->stream));
    return VH_OK;
}

// ---------------------------------------------------------------------------
// model dump / checkpoint
// ---------------------------------------------------------------------------
// SDFRenderer::printSDFdata (SDFRenderer.cpp:71-110), the reference's only on-disk artefact:
// the occupied count, then per compact entry "pos / ptr / offset" and 512 sdf values with 4
// decimals.  Faithful to a quirk of the original: the 512 values printed for entry i are voxels
// [512*i, 512*i+512) of the volume (it reads the first count*512 voxels, :85-87), NOT the block
// the entry's ptr names.
extern "C" int vh_dump_sdf_text(vh_context *c, const char *path)
{
    if (!c || !path) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    vh_counters k;
    int rc = vh_get_counters(c, &k);
    if (rc != VH_OK) return rc;
    const size_t n = (size_t)(k.occupied > 0 ? k.occupied : 0);
    std::vector<VoxelEntry> entries(n);
    const size_t nvox = std::min(n * kBlockVoxels, (size_t)c->params.numVoxelBlocks * kBlockVoxels);
    std::vector<Voxel> vox(n * kBlockVoxels, Voxel{0.0f, 0.0f});
    if (n) {
        if ((rc = vh_download(c, VH_BUF_COMPACT, entries.data(), n * sizeof(VoxelEntry))) != VH_OK) return rc;
        if ((rc = vh_download(c, VH_BUF_SDF_BLOCKS, vox.data(), nvox * sizeof(Voxel))) != VH_OK) return rc;
    }
    FILE *f = std::fopen(path, "w");
    if (!f) return fail(VH_ERR_INVALID_ARGUMENT, "cannot open the dump file");
    std::fprintf(f, "numOccupiedBlocks from GL :%zu\n", n);                                  // :95
    std::fprintf(f, "\nSDFs \n\n");                                                           // :100
    for (size_t i = 0; i < n; ++i) {
        const VoxelEntry &e = entries[i];
        std::fprintf(f, "%zu) : pos : (%d, %d, %d) ptr = %d offset = %d\n", i, e.pos[0], e.pos[1], e.pos[2], e.ptr,
                     e.offset);                                                               // :102-103
        for (int j = 0; j < kBlockVoxels; ++j) std::fprintf(f, "%.4f\t", vox[i * kBlockVoxels + j].sdf);   // :104-106
        std::fprintf(f, "\n\n\n");
    }
    std::fclose(f);
    return VH_OK;
}

// Binary snapshot: header, hash table, heap, then the 4 KiB block of every allocated entry in
// table order.  Enough to continue fusing after vh_load_snapshot as if never interrupted.
struct SnapshotHeader {
    char magic[8];                 // "VHSNAP01"
    HashTableParams params;
    int32_t width, height, semantics;
    uint32_t bucketLo, bucketHi;
    int32_t heapCounter;
    uint32_t allocatedTotal, heapExhausted, epoch;
    uint64_t numEntries, numAllocated;
    float proj[9];
};

extern "C" int vh_save_snapshot(vh_context *c, const char *path)
{
    if (!c || !path) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    vh_counters k;
    int rc = vh_get_counters(c, &k);
    if (rc != VH_OK) return rc;
    std::vector<VoxelEntry> table(c->numEntries);
    std::vector<uint32_t> heap(c->params.numVoxelBlocks);
    if ((rc = vh_download(c, VH_BUF_HASH_TABLE, table.data(), table.size() * sizeof(VoxelEntry))) != VH_OK) return rc;
    if ((rc = vh_download(c, VH_BUF_HEAP, heap.data(), heap.size() * sizeof(uint32_t))) != VH_OK) return rc;
    SnapshotHeader h{};
    std::memcpy(h.magic, "VHSNAP01", 8);
    h.params = c->params;
    h.width = c->fp.width; h.height = c->fp.height; h.semantics = c->fp.semantics;
    h.bucketLo = c->fp.bucketLo; h.bucketHi = c->fp.bucketHi;
    h.heapCounter = k.heap_counter; h.allocatedTotal = k.allocated_total; h.heapExhausted = k.heap_exhausted;
    h.epoch = c->fp.epoch;
    h.numEntries = c->numEntries;
    std::memcpy(h.proj, c->fp.proj, sizeof h.proj);
    for (const VoxelEntry &e : table) h.numAllocated += e.ptr != VH_FREE_BLOCK;
    FILE *f = std::fopen(path, "wb");
    if (!f) return fail(VH_ERR_INVALID_ARGUMENT, "cannot open the snapshot file");
    bool ok = std::fwrite(&h, sizeof h, 1, f) == 1;
    ok = ok && std::fwrite(table.data(), sizeof(VoxelEntry), table.size(), f) == table.size();
    ok = ok && std::fwrite(heap.data(), sizeof(uint32_t), heap.size(), f) == heap.size();
    DeviceGuard guard(c->device);
    std::vector<Voxel> block(kBlockVoxels);
    for (const VoxelEntry &e : table) {
        if (e.ptr == VH_FREE_BLOCK || !ok) continue;
        if (hipMemcpy(block.data(), c->dp.blocks + e.ptr, sizeof(Voxel) * kBlockVoxels, hipMemcpyDeviceToHost) !=
            hipSuccess) { ok = false; break; }
        ok = std::fwrite(block.data(), sizeof(Voxel), kBlockVoxels, f) == (size_t)kBlockVoxels;
    }
    std::fclose(f);
    return ok ? VH_OK : fail(VH_ERR_HIP, "snapshot write failed");
}

extern "C" int vh_load_snapshot(vh_context *c, const char *path)
{
    if (!c || !path) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    FILE *f = std::fopen(path, "rb");
    if (!f) return fail(VH_ERR_INVALID_ARGUMENT, "cannot open the snapshot file");
    SnapshotHeader h;
    bool ok = std::fread(&h, sizeof h, 1, f) == 1 && std::memcmp(h.magic, "VHSNAP01", 8) == 0;
    ok = ok && h.numEntries == c->numEntries && h.params.numVoxelBlocks == c->params.numVoxelBlocks &&
         h.params.numBuckets == c->params.numBuckets && h.params.bucketSize == c->params.bucketSize &&
         h.bucketLo == c->fp.bucketLo && h.bucketHi == c->fp.bucketHi && h.width == c->fp.width &&
         h.height == c->fp.height;
    if (!ok) { std::fclose(f); return fail(VH_ERR_INVALID_ARGUMENT, "snapshot does not match this context"); }
    std::vector<VoxelEntry> table(c->numEntries);
    std::vector<uint32_t> heap(c->params.numVoxelBlocks);
    ok = std::fread(table.data(), sizeof(VoxelEntry), table.size(), f) == table.size() &&
         std::fread(heap.data(), sizeof(uint32_t), heap.size(), f) == heap.size();
    DeviceGuard guard(c->device);
    hipError_t e = hipStreamSynchronize(c->stream);
    const size_t words = ((size_t)c->ownedBuckets + 31) / 32;
    std::vector<uint32_t> bits(words, 0u), macro(kMacroBits / 32, 0u);
    std::vector<Voxel> block(kBlockVoxels);
    if (ok && e == hipSuccess)
        e = hipMemset(c->dp.blocks, 0, sizeof(Voxel) * (size_t)c->params.numVoxelBlocks * kBlockVoxels);
    for (size_t i = 0; ok && e == hipSuccess && i < table.size(); ++i) {
        if (table[i].ptr == VH_FREE_BLOCK) continue;
        const size_t bucket = i / c->params.bucketSize;
        bits[bucket >> 5] |= 1u << (bucket & 31);
        const uint32_t hm = ((((uint32_t)(table[i].pos[0] >> 2)) * 73856093u) ^ (((uint32_t)(table[i].pos[1] >> 2)) * 19349669u) ^
                             (((uint32_t)(table[i].pos[2] >> 2)) * 83492791u)) & (kMacroBits - 1u);
        macro[hm >> 5] |= 1u << (hm & 31);
        ok = std::fread(block.data(), sizeof(Voxel), kBlockVoxels, f) == (size_t)kBlockVoxels &&
             (uint64_t)table[i].ptr + kBlockVoxels <= (uint64_t)c->params.numVoxelBlocks * kBlockVoxels;
        if (ok) e = hipMemcpy(c->dp.blocks + table[i].ptr, block.data(), sizeof(Voxel) * kBlockVoxels, hipMemcpyHostToDevice);
    }
    std::fclose(f);
    if (!ok) return fail(VH_ERR_INVALID_ARGUMENT, "snapshot is truncated or corrupt");
    int32_t counters[kNumCounters] = {0};
    counters[kHeapCounter] = h.heapCounter;
    counters[kAllocatedTotal] = (int32_t)h.allocatedTotal;
    counters[kHeapExhausted] = (int32_t)h.heapExhausted;
    if (e == hipSuccess) e = hipMemcpy(c->dp.table, table.data(), sizeof(VoxelEntry) * table.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(c->dp.heap, heap.data(), sizeof(uint32_t) * heap.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(c->dp.bucketBits, bits.data(), sizeof(uint32_t) * words, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(c->dp.macroBits, macro.data(), kMacroBits / 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(c->dp.claim, 0, sizeof(unsigned long long) * (size_t)c->ownedBuckets);
    if (e == hipSuccess) e = hipMemcpy(c->dp.counters, counters, sizeof counters, hipMemcpyHostToDevice);
    if (e != hipSuccess) return fail(VH_ERR_HIP, "snapshot upload", e);
    c->params = h.params;
    std::memcpy(c->fp.T, h.params.global_transform, sizeof c->fp.T);
    std::memcpy(c->fp.Tinv, h.params.inv_global_transform, sizeof c->fp.Tinv);
    std::memcpy(c->fp.proj, h.proj, sizeof h.proj);
    c->fp.semantics = h.semantics;
    c->fp.epoch = 0;                 // the claim words were cleared: any epoch >= 1 is fresh
    c->fusedParity = 0;
    c->compactArmed = false;
    c->occupiedCounter = kCompactCount;
    return VH_OK;
}

extern "C" int vh_set_option(vh_context *c, const char *name, int value)
{
    if (!c || !name) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    if (std::strcmp(name, "flatten_variant") == 0) { c->flattenVariant = value; return VH_OK; }
    if (std::strcmp(name, "integrate_grid") == 0 && value > 0) { c->integrateGrid = value; return VH_OK; }
    if (std::strcmp(name, "fused_frame") == 0) { c->fusedFrame = value; return VH_OK; }
    if (std::strcmp(name, "raycast_patch") == 0) { c->raycastPatch = value; return VH_OK; }
    if (std::strcmp(name, "raycast_xcd") == 0) { c->raycastXcd = value; return VH_OK; }
    if (std::strcmp(name, "packet_format") == 0 && (value == VH_PACKET_F32 || value == VH_PACKET_U16)) {
        c->packetFormat = value;
        return VH_OK;
    }
    if (std::strcmp(name, "persistent_blocks") == 0 && value > 0) { c->persistentBlocks = value; return VH_OK; }
    if (std::strcmp(name, "commit_blocks") == 0 && value > 0) { c->commitBlocks = value; return VH_OK; }
    return fail(VH_ERR_INVALID_ARGUMENT, "unknown option");
}

extern "C" int vh_set_profiling(vh_context *c, int enabled)
{
    if (!c) return fail(VH_ERR_INVALID_ARGUMENT, "null context");
    c->profiling = enabled != 0;
    return VH_OK;
}

extern "C" int vh_get_kernel_times(vh_context *c, vh_kernel_times *out, int reset)
{
    if (!c || !out) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    DeviceGuard guard(c->device);
    VH_HIP(hipStreamSynchronize(c->stream));
    for (auto &t : c->timed) {
        float ms = 0;
        VH_HIP(hipEventElapsedTime(&ms, t.start, t.stop));
        switch (t.phase) {
            case kPhaseClaim: c->times.alloc_claim_ms += ms; break;
            case kPhaseCommit: c->times.alloc_commit_ms += ms; break;
            case kPhaseFlatten: c->times.flatten_ms += ms; break;
            case kPhaseIntegrate: c->times.integrate_ms += ms; break;
            case kPhaseRaycast: c->times.raycast_ms += ms; c->times.raycast_launches += 1; break;
            case kPhaseFrameScanClaim: c->times.frame_scan_claim_ms += ms; break;
            case kPhaseFrameCommitIntegrate: c->times.frame_commit_integrate_ms += ms; break;
            case kPhaseViewExport: c->times.view_export_ms += ms; break;
            case kPhaseViewImport: c->times.view_import_ms += ms; break;
            case kPhaseGc: c->times.gc_ms += ms; break;
            case kPhaseRaycastBounds: c->times.raycast_ms += ms; break;      // vh_render_blocks: counted with the render work
            default: break;
        }
    }
    c->times.launches += c->profiledFrames;
    c->profiledFrames = 0;
    drop_events(c);
    *out = c->times;
    if (reset) c->times = vh_kernel_times{};
    return VH_OK;
}

// test hook: scalar helpers evaluated on the device (8 int32 per point:
// block x,y,z, hash, inFrustum, screen x,y, f2i_rz(w))
extern "C" int vh_debug_eval(vh_context *c, const vh_float4 *d_points, int32_t n, int32_t *d_out)
{
    if (!c || !d_points || !d_out || n < 0) return fail(VH_ERR_INVALID_ARGUMENT, "bad argument");
    DeviceGuard guard(c->device);
    if (n == 0) return VH_OK;
    debug_eval_kernel<<<grid_for((size_t)n, 256), 256, 0, c->stream>>>(c->fp, reinterpret_cast<const float4 *>(d_points), n,
                                                                       d_out);
    VH_HIP(hipGetLastError());
    return VH_OK;
}

// ---------------------------------------------------------------------------
// depth pre-processing (CameraTrackingUtils.cu:115-120, 218-222)
// ---------------------------------------------------------------------------
extern "C" int vh_preprocess(const uint16_t *d_depth, const float k_inv[9], int32_t width, int32_t height,
                             vh_float4 *d_positions, vh_float4 *d_normals, void *hip_stream)
{
    if (!d_depth || !k_inv || !d_positions || !d_normals || width <= 0 || height <= 0 ||
        (uint64_t)width * height > (1u << 24))
        return fail(VH_ERR_INVALID_ARGUMENT, "bad argument");
    Mat3 k;
    std::memcpy(k.m, k_inv, sizeof k.m);
    preprocess_kernel<<<grid_for((size_t)width * height, 256), 256, 0, (hipStream_t)hip_stream>>>(
        d_depth, k, width, height, reinterpret_cast<float4 *>(d_positions), reinterpret_cast<float4 *>(d_normals));
    VH_HIP(hipGetLastError());
    return VH_OK;
}

static float g_k_inv[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
static float g_k[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};

extern "C" bool SetCameraIntrinsic(const float *intrinsic, const float *invIntrinsic)
{
    if (!invIntrinsic) return false;
    if (intrinsic) std::memcpy(g_k, intrinsic, sizeof g_k);          // K feeds computeCorrespondences
    std::memcpy(g_k_inv, invIntrinsic, sizeof g_k_inv);
    return true;
}

extern "C" void preProcess(vh_float4 *positions, vh_float4 *normals, const uint16_t *depth)
{
    // 640x480 and the default stream, like the reference (CameraTrackingUtils.cu:28-36,115-120)
    int rc = vh_preprocess(depth, g_k_inv, 640, 480, positions, normals, nullptr);
    if (rc == VH_OK && hipDeviceSynchronize() != hipSuccess) rc = VH_ERR_HIP;
    if (rc != VH_OK) {
        std::fprintf(stderr, "voxelhash: preProcess failed: %s (%s)\n", vh_error_string(rc), vh_last_error());
        std::exit(EXIT_FAILURE);
    }
}

// ---------------------------------------------------------------------------
// drop-in names (VoxelUtils.h:5-13) on a process-global context
// ---------------------------------------------------------------------------
static vh_context *g_default = nullptr;
static HashTableParams g_default_params;
static bool g_have_params = false;

[[noreturn]] static void die(const char *where, int rc)
{
    // checkCudaErrors convention, helper_cuda.h:966-977
    std::fprintf(stderr, "voxelhash: %s failed: %s (%s)\n", where, vh_error_string(rc), vh_last_error());
    std::exit(EXIT_FAILURE);
}

extern "C" vh_context *vh_default_context(void) { return g_default; }

extern "C" void updateConstantHashTableParams(const HashTableParams *params)
{
    // VoxelUtils.cu:87-91.  There is no __constant__ copy to refresh: kernels
    // receive the frame parameters by value.  The pose and the occupied count
    // are taken over.
    if (!params) die("updateConstantHashTableParams", VH_ERR_INVALID_ARGUMENT);
    g_default_params = *params;
    g_have_params = true;
    if (g_default) {
        std::memcpy(g_default->fp.T, params->global_transform, sizeof g_default->fp.T);
        std::memcpy(g_default->fp.Tinv, params->inv_global_transform, sizeof g_default->fp.Tinv);
        std::memcpy(g_default->params.global_transform, params->global_transform, sizeof g_default->fp.T);
        std::memcpy(g_default->params.inv_global_transform, params->inv_global_transform, sizeof g_default->fp.T);
        g_default->params.numOccupiedBlocks = params->numOccupiedBlocks;
    }
}

extern "C" void deviceAllocate(const HashTableParams *params)
{
    if (!params) die("deviceAllocate", VH_ERR_INVALID_ARGUMENT);
    if (g_default) { vh_destroy(g_default); g_default = nullptr; }
    vh_config cfg;
    cfg.params = *params;
    cfg.width = 640;          // common.h:17-18
    cfg.height = 480;
    cfg.semantics = VH_SEM_REFERENCE;
    cfg.device = -1;
    if (const char *s = std::getenv("VOXELHASH_SEMANTICS"))
        if (std::strcmp(s, "pinhole") == 0) cfg.semantics = VH_SEM_PINHOLE;
    int rc = vh_create(&cfg, &g_default);
    if (rc != VH_OK) die("deviceAllocate", rc);
}

extern "C" void deviceFree(void)
{
    if (g_default) { vh_destroy(g_default); g_default = nullptr; }
}

extern "C" void resetHashTableMutexes(const HashTableParams *params)
{
    (void)params;
    if (!g_default) die("resetHashTableMutexes", VH_ERR_NOT_INITIALISED);
    int rc = vh_reset_mutexes(g_default);
    if (rc != VH_OK) die("resetHashTableMutexes", rc);
}

extern "C" void allocBlocks(const vh_float4 *verts, const vh_float4 *normals)
{
    if (!g_default) die("allocBlocks", VH_ERR_NOT_INITIALISED);
    int rc = vh_alloc_blocks(g_default, verts, normals);
    if (rc == VH_OK) rc = vh_synchronize(g_default);     // the reference syncs after the launch (:715)
    if (rc != VH_OK) die("allocBlocks", rc);
}

extern "C" int flattenIntoBuffer(const HashTableParams *params)
{
    (void)params;
    if (!g_default) die("flattenIntoBuffer", VH_ERR_NOT_INITIALISED);
    int32_t n = 0;
    int rc = vh_flatten(g_default, &n);
    if (rc != VH_OK) die("flattenIntoBuffer", rc);
    return n;
}

extern "C" void calculateKinectProjectionMatrix(void)
{
    if (!g_default) die("calculateKinectProjectionMatrix", VH_ERR_NOT_INITIALISED);
    default_projection(g_default);                       // VoxelUtils.cu:224-231
}

extern "C" void integrateDepthMap(const HashTableParams *params, const vh_float4 *verts)
{
    if (!g_default) die("integrateDepthMap", VH_ERR_NOT_INITIALISED);
    if (params && params->numOccupiedBlocks == 0) return;          // :848
    int rc = vh_integrate_depth_map(g_default, verts);
    if (rc == VH_OK) rc = vh_synchronize(g_default);               // :850
    if (rc != VH_OK) die("integrateDepthMap", rc);
}

#include "vh_api_icp.hip"
