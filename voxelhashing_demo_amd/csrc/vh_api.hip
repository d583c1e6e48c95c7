// vh_api.hip -- the C-ABI of libvoxelhash_hip.so (include/voxelhash.h): context
// lifecycle, per-frame orchestration (SDF_Hashtable.cpp:11-40 without its four
// device syncs and two D2H reads), and the reference's drop-in names.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
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
// tracing: roctx ranges around the entry points (tracing row)
// ---------------------------------------------------------------------------
// Off unless the environment has VOXELHASH_ROCTX=1 (one getenv at the first entry point): libroctx64 is then loaded at run time
// (no link dependency) and every frame / raycast / exchange / ICP entry point pushes a range named after itself, which
// `rocprofv3 --marker-trace` shows beside the kernels it launched.
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx()
    {
        const char *e = std::getenv("VOXELHASH_ROCTX");
        if (!e || std::atoi(e) == 0) return;
        void *lib = nullptr;
        // (rocprofv3 = rocprofiler-sdk records the ranges of ITS roctx library; libroctx64 is roctracer's, for the older tools)
        for (const char *n : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"})
            if ((lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL)) != nullptr) break;
        if (!lib) return;
        push = reinterpret_cast<int (*)(const char *)>(dlsym(lib, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(lib, "roctxRangePop"));
        if (!push || !pop) push = nullptr;
    }
};
static const Roctx &roctx() { static const Roctx r; return r; }
struct RoctxRange {
    bool on;
    explicit RoctxRange(const char *name) : on(roctx().push != nullptr) { if (on) roctx().push(name); }
    ~RoctxRange() { if (on) roctx().pop(); }
    RoctxRange(const RoctxRange &) = delete;
    RoctxRange &operator=(const RoctxRange &) = delete;
};
#define VH_TRACE(name) RoctxRange vh_trace_range_(name)

// ---------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------
// Profiling attaches a start/stop event pair to each dispatch (hipExtLaunchKernelGGL):
// the pair carries the begin/end timestamps of that kernel alone, the same
// quantity rocprofv3 --kernel-trace reports, without the inter-kernel gaps a
// hipEventRecord pair would add.
enum Phase : int {
    kPhaseClaim = 0, kPhaseCommit, kPhaseFlatten, kPhaseIntegrate, kPhaseRaycast,
    kPhaseFrameScanClaim, kPhaseFrameCommitIntegrate, kPhaseViewExport, kPhaseViewImport, kPhaseGc, kPhaseRaycastBounds, kPhaseFramePipelined, kNumPhases
};

struct TimedLaunch {
    int phase;
    hipEvent_t start, stop;
};

struct MultiPending {
    bool active = false;
    uint32_t epochOld = 0;
    int32_t doneTag = 0;                   // what that frame's commit phase publishes (overflow list)
    const float *packetsOld = nullptr;     // the caller's packets of that frame: valid until the half has been launched
    size_t packetStride = 0;
    int32_t numCams = 0;
    int packetFormat = 0;
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
    std::vector<std::pair<hipEvent_t, hipEvent_t>> eventPool;   // idle event pairs, reused by launch()
    uint64_t profiledFrames = 0;
    vh_kernel_times times{};
    int integrateGrid = 2048;
    void *raycastStamps = nullptr; // diagnostics: device buffer of 4 uint64 per wave (vh_debug_set_raycast_stamps)
    int raycastBeam = 3;           // DDA: 2 = cooperative form (one block list per wave), 1 = per-lane walk behind the beam front end, 0 = per-lane walk,
                                   // 3 = by the view: cooperative when 64 half-block slabs span [t_min, t_max], else 1 (option "raycast_beam")
    int raycastMode = VH_RAYCAST_DDA;   // option "raycast_mode": voxel DDA (raycastSDF.frag:121-177) or the fixed-step march
    int packetFormat = VH_PACKET_F32;   // what vh_integrate_packets / vh_apply_frames_batch read
    int fusedFrame = 1;            // vh_integrate as two launches (0: the four step kernels)
    int commitBlocks = 128;        // workgroups serving candidates in the fused second launch
    int fusedParity = 0;           // which of the two per-frame counter sets the next fused frame uses
    bool compactArmed = false;     // alloc_commit has zeroed the compact counter and no flatten ran since
    // WalkKind (option "flatten_variant"): 3 = the reference's walk over every VoxelEntry (flattenKernel, VoxelUtils.cu:719-749),
    // 4 = the walk over the bucket-occupancy bitmap and the non-empty buckets (same compact set, 2-11 x the frames/s): the default
    // since round 6.  The reference's walk stays selectable, and it is what bench.py's `value` measures (20*N).
    int flattenVariant = 4;
    uint32_t candAllocated = 0;    // records the candidate buffer holds (dp.candCapacity <= this)
    uint32_t allocEpoch = 0;       // lock epoch (epochTotal) of the last allocBlocks (overflow list: one per epoch)
    uint32_t epochTotal = 0;       // lock epochs since creation (fp.epoch is the 9-bit epoch of the claim words)
    // pipelined frames (option "pipeline", vh_integrate_batch; vh_frame.hip)
    float *fusedPlane = nullptr;   // packed camera-z plane launch 1 of the two-launch frame leaves for launch 2 (large images)
    int pipeline = 0;
    int pipeIntegrateGrid = 512;   // workgroups of the deferred TSDF update inside a pipelined launch
    bool pipePending = false;      // the commit + TSDF update of the last frame are still to be launched
    FrameParams pipeFp;            // that frame's parameters
    int32_t pipeDoneTag = 0;       // ... and its tag (lock epochs since creation): what its commit phase publishes (overflow list)
    int pipeSet = 0;               // counter set its claim / walk filled
    int pipeParity = 0;            // which of the two buffer sets it used
    int pipeSensor = 0;            // its private depth copy: 0 = float camera-z plane, 1 = uint16 sensor image
    float pipeK[4] = {0, 0, 0, 0}; // K_inv row 2 and the depth unit of a sensor frame
    unsigned long long *claimBuf[2] = {nullptr, nullptr};
    int4 *candBuf[2] = {nullptr, nullptr};
    VoxelEntry *compactBuf[2] = {nullptr, nullptr};
    uint32_t *claimFilter = nullptr;       // pipelined frames: three claim filters of kPendFilterWords words (vh_alloc.hip: pend_maybe)
    uint32_t *maskBuf2 = nullptr;          // pipelined multi-camera frames: the camera masks of the second compact buffer
    int genFramesPerLaunch = 4;            // option "gen_frames_per_launch": frames of a batch one key-generation launch takes (1..8)
    uint32_t spinLimit = 0;                // option "spin_limit": polls a workgroup of a serialised pipelined launch waits for the pending commit phase (0: kSpinLimitDefault)
    int pipelineOverflow = 1;              // option "pipeline_overflow": one-launch (serialised) frames with the overflow list: 0 never, 1 by the launch's size (default), 2 always
    bool serialQueued = false;             // a serialised launch has been queued since the host last looked at kSpinTimeouts (check_spin_timeouts)
    uint32_t spinSeen = 0;                 // ... and what the counter read then
    bool serialFallback = false;           // a serialised launch has timed out (vh_counters.spin_timeouts): overflow-list frames take two launches from now on
    int debugSkipRoles = 0;                // diagnostics: roles of the pipelined launch that return at once (timing only; the model is wrong)
    int pipelineShards = 1;                // option "pipeline_shards": vh_apply_frames_batch runs a batch of B multi-camera frames as B + 1 launches (1) or B (2: the last frame's half stays pending across calls)
    MultiPending multiPend;                // the multi-camera frame whose commit + TSDF update have not been launched yet
    void *multiFirstEvent = nullptr;       // hipEvent_t recorded behind the first launch of every vh_apply_frames_batch (vh_dist: the previous batch's packets are free)
    VoxelEntry *compactHome = nullptr;     // the compact buffer of creation: what PtrContainer names, where settle() leaves the dense list
    float *planeBuf[2] = {nullptr, nullptr};
    uint16_t *rawBuf[2] = {nullptr, nullptr};
    int occupiedCounter = kCompactCount;   // which device counter holds the occupied count of the last frame
    // raycast over shards
    void *viewSet = nullptr;               // vh_export_views_fixed: the prepared views (ViewSet) in device memory
    int32_t *viewLists = nullptr;          // export: selected entry indices, [views][capacity]
    size_t viewListsSize = 0;              // in int32
    int32_t *blockList = nullptr;          // vh_render_blocks: two counter words (4 ints) + the records of the allocated blocks
    size_t blockCapacity = 0;              // records blockList has room for
    int foldA = -1, foldB = -1, foldNew = -1;   // counters of the last frame's two-ended compact list while it is unfolded (foldA < 0: dense)
    int blockParity = 0;                   // which counter word the next vh_render_blocks appends through
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
        case VH_ERR_TIMEOUT: return "workgroups of a launch that waits on itself gave up waiting (serialised frame: frames have lost work; vh_icp_align: call again)";
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

static int check_spin_timeouts(vh_context *c);  // vh_api_model.hip: behind a host synchronisation, VH_ERR_TIMEOUT if a serialised launch gave up
static int flush_pending(vh_context *c);       // vh_api_frame.hip: launches a pipelined frame's deferred half
static int flush_single_pending(vh_context *c);
static int flush_multi_pending(vh_context *c);     // vh_api_shard.hip: the same for a multi-camera frame (pipeline_shards 2)
static int settle(vh_context *c);              // ... and folds a two-ended compact list into the dense one (observers)

static int free_buffers(vh_context *c)
{
    // the second buffer set of the pipelined frames (set 0 aliases dp.claim / dp.candidates / dp.compact of creation)
    for (int i = 0; i < 2; ++i) {
        if (c->claimBuf[i] && c->claimBuf[i] != c->dp.claim) (void)hipFree(c->claimBuf[i]);
        if (c->candBuf[i] && c->candBuf[i] != c->dp.candidates) (void)hipFree(c->candBuf[i]);
        if (c->compactBuf[i] && c->compactBuf[i] != c->dp.compact) (void)hipFree(c->compactBuf[i]);
        if (c->planeBuf[i]) (void)hipFree(c->planeBuf[i]);
        if (c->rawBuf[i]) (void)hipFree(c->rawBuf[i]);
        c->claimBuf[i] = nullptr; c->candBuf[i] = nullptr; c->compactBuf[i] = nullptr;
        c->planeBuf[i] = nullptr; c->rawBuf[i] = nullptr;
    }
    if (c->dp.heap) (void)hipFree(c->dp.heap);
    if (c->dp.table) (void)hipFree(c->dp.table);
    if (c->dp.compact) (void)hipFree(c->dp.compact);
    if (c->dp.claim) (void)hipFree(c->dp.claim);
    if (c->dp.blocks) (void)hipFree(c->dp.blocks);
    if (c->dp.counters) (void)hipFree(c->dp.counters);
    if (c->dp.candidates) (void)hipFree(c->dp.candidates);
    if (c->dp.candTarget) (void)hipFree(c->dp.candTarget);
    if (c->dp.gcMarks) (void)hipFree(c->dp.gcMarks);
    if (c->dp.compactMask) (void)hipFree(c->dp.compactMask);
    if (c->dp.bucketBits) (void)hipFree(c->dp.bucketBits);
    if (c->dp.macroBits) (void)hipFree(c->dp.macroBits);
    if (c->fusedPlane) (void)hipFree(c->fusedPlane);
    c->fusedPlane = nullptr;
    if (c->maskBuf2) (void)hipFree(c->maskBuf2);
    c->maskBuf2 = nullptr;
    if (c->claimFilter) (void)hipFree(c->claimFilter);
    c->claimFilter = nullptr;
    if (c->viewSet) (void)hipFree(c->viewSet);
    c->viewSet = nullptr;
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
    fp.flags = 0;
    fp.listSize = p.attachedLinkedListSize;
    fp.truncScale = p.truncScale;
    fp.weightSample = p.integrationWeightSample;
    default_projection(c);

    c->ownedBuckets = hi - lo;
    c->numEntries = (size_t)c->ownedBuckets * p.bucketSize;
    // A table beyond the 256 MiB Infinity Cache is walked with non-temporal loads: a pure read of 419 MB runs
    // at 7.0 instead of 6.0 TB/s that way (tools/micro/membw.hip) and the voxel blocks stay cached across
    // frames (C3: launch 1 76.2 -> 70.2 us, launch 2 13.5 -> 11.7 us); a resident table (C2, 105 MB) loses
    // 2 % with them.  Option "walk_nt" overrides.
    if (c->numEntries * sizeof(VoxelEntry) > ((size_t)256 << 20)) fp.flags |= kFlagWalkNt;
    fp.flags |= kFlagWalkShort;          // 4 entries per lane in the frame's walk (option "walk_entries": 4 | 8)
    const size_t npix = (size_t)cfg->width * cfg->height;
    DevPtrs &dp = c->dp;
    dp = DevPtrs{};
    dp.candCapacity = c->candAllocated = (uint32_t)std::min<size_t>(npix, kMaxCandidates);

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
    VH_ALLOC(dp.table, sizeof(VoxelEntry) * c->numEntries);
    VH_ALLOC(dp.compact, sizeof(VoxelEntry) * c->numEntries);
    VH_ALLOC(dp.claim, sizeof(unsigned long long) * (size_t)c->ownedBuckets);
    VH_ALLOC(dp.blocks, sizeof(Voxel) * (size_t)p.numVoxelBlocks * kBlockVoxels);
    VH_ALLOC(dp.counters, sizeof(int32_t) * kNumCounters);
    VH_ALLOC(dp.candidates, sizeof(int4) * npix);
    VH_ALLOC(dp.candTarget, sizeof(uint32_t) * npix);
    VH_ALLOC(dp.gcMarks, sizeof(uint32_t) * ((c->numEntries + 31) / 32));
    VH_ALLOC(dp.compactMask, sizeof(uint32_t) * c->numEntries);
    VH_ALLOC(dp.bucketBits, sizeof(uint32_t) * (((size_t)c->ownedBuckets + 31) / 32 + 1));   // (+1: the DDA raycast reads it with 8-byte loads)
    VH_ALLOC(dp.macroBits, kMacroBits / 8);
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
    if (e == hipSuccess) e = hipMemsetAsync(dp.gcMarks, 0, sizeof(uint32_t) * ((c->numEntries + 31) / 32), s);
    if (e == hipSuccess) e = hipMemsetAsync(dp.blocks, 0, sizeof(Voxel) * (size_t)p.numVoxelBlocks * kBlockVoxels, s);
    c->compactHome = dp.compact;
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
    for (auto &p : c->eventPool) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    c->eventPool.clear();
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
    if (c->pipePending) {            // the deferred half of the last frame belongs on the stream that frame ran on
        DeviceGuard guard(c->device);
        const int rc = flush_pending(c);
        if (rc != VH_OK) return rc;
    }
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

// The candidate list holds one record per contender of a lock epoch.  A single-camera frame has at
// most one per pixel and band sample after the wave-level collapse (W*H records are allocated; more
// is counted in vh_counters.cand_overflow); a multi-camera frame can bring up to
// num_bins*(capacity-1) keys, for which the list is grown here (synchronises when it grows).
static int ensure_candidates(vh_context *c, size_t need)
{
    if (need > kMaxCandidates) need = kMaxCandidates;        // a claim word names its record with 19 bits; more is counted as overflow
    if (need <= c->candAllocated) return VH_OK;
    int rc = flush_pending(c);
    if (rc != VH_OK) return rc;
    VH_HIP(hipStreamSynchronize(c->stream));
    int4 *fresh = nullptr, *fresh2 = nullptr;
    uint32_t *freshTarget = nullptr;
    const bool two = c->candBuf[0] != nullptr;               // the pipelined frames' second list
    hipError_t e = hipMalloc((void **)&fresh, sizeof(int4) * need);
    if (e == hipSuccess) e = hipMalloc((void **)&freshTarget, sizeof(uint32_t) * need);
    if (e == hipSuccess && two) e = hipMalloc((void **)&fresh2, sizeof(int4) * need);
    if (e != hipSuccess) {
        if (fresh) (void)hipFree(fresh);
        if (freshTarget) (void)hipFree(freshTarget);
        return fail(VH_ERR_OUT_OF_MEMORY, "hipMalloc candidate list", e);
    }
    if (two) {
        (void)hipFree(c->candBuf[0]);
        (void)hipFree(c->candBuf[1]);
        c->candBuf[0] = fresh;
        c->candBuf[1] = fresh2;
    } else {
        (void)hipFree(c->dp.candidates);
    }
    (void)hipFree(c->dp.candTarget);
    c->dp.candidates = fresh;
    c->dp.candTarget = freshTarget;
    c->dp.candCapacity = c->candAllocated = (uint32_t)need;
    return VH_OK;
}

// the rest of the C-ABI, by area (one translation unit)
#include "vh_api_frame.hip"
#include "vh_api_shard.hip"
#include "vh_api_model.hip"
#include "vh_api_dropin.hip"
#include "vh_api_icp.hip"
#include "vh_api_dist.hip"
