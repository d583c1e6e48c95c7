// vh_device.h -- device-side records and scalar helpers of the voxel-hashing path
// (gfx950 only).  Every helper states the reference line it reproduces; the file
// is compiled with -ffp-contract=off because the reference is built with
// nvcc -fmad=false (CMakeLists.txt:23): no multiply-add may be fused.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/voxelhash.h"

static_assert(sizeof(Voxel) == 8, "Voxel must stay 8 bytes (VoxelDataStructures.h:12-17)");
static_assert(sizeof(VoxelEntry) == 20, "VoxelEntry must stay 20 bytes (VoxelDataStructures.h:20-26)");
static_assert(sizeof(HashTableParams) == 176, "HashTableParams must stay 176 bytes");

namespace vh {

constexpr int kWave = 64;
constexpr int kBlockVoxels = 512;          // 8^3, the *512 literals of VoxelUtils.cu:449
constexpr int kEntryDwords = 5;            // 20-byte VoxelEntry

// device counters (one int32 array per context)
enum Counter : int {
    kHeapCounter = 0,      // index of the top free heap slot (VoxelUtils.cu:207)
    kCompactCount = 1,     // d_compactifiedHashCounter
    kCandCount = 2,        // contenders recorded by the running allocBlocks
    kAllocatedTotal = 3,
    kHeapExhausted = 4,
    kLastCandidates = 5,
    kCommitTicket = 6,
    kBinOverflow = 7,      // a received key bin carried more keys than its capacity
    // fused frame: per-frame counters, double-buffered by epoch parity so that no workgroup
    // has to wait for all others before they can be re-armed (frame f's second launch
    // clears the set frame f+1 will use)
    kScanCount = 8,        // [2] entries found by the table walk: list A, growing up from compact[0]
    kScanCountB = 40,      // [2] ... list B, growing down from the last compact entry (its own cache line, see CompactOut)
    kNewCount = 10,        // [2] entries inserted (and appended) by the commit phase
    kFusedCand = 12,       // [2] contenders recorded by the claim phase
    // deletion / garbage collection (vh_gc.hip)
    kGcBuckets = 14,       // buckets on the sweep list of the running collection
    kGcFreed = 15,         // blocks on its freed list
    kFreedTotal = 16,      // blocks returned to the heap since creation
    kLastFreed = 17,       // ... by the last vh_delete_blocks / vh_garbage_collect
    kCandOverflow = 18,    // contenders dropped because the candidate list was full (never cleared)
    kSpinTimeouts = 19,    // workgroups of a serialised pipelined launch that gave up waiting for the pending frame's commit phase (never cleared)
    // pipelined frames (vh_integrate_batch): three rotating sets -- the launch of frame i+1 fills set
    // (i+1)%3 (claim / walk), consumes set i%3 (commit / integrate of frame i) and clears set (i+2)%3
    // Each set has a 128-byte line of its own (kPipeSetStride ints apart; the host hands the kernel set * stride):
    // the words of the set a launch only READS (every workgroup looks at the old frame's candidate count,
    // free-block count and winner count) then do not share a cache line with the words the same launch
    // increments atomically.
    kPipeScan = 64,
    kPipeNew = 65,
    kPipeCand = 66,
    kPipeHeapFree = 67,    // free blocks on the heap when the launch that consumes the set began (written by the
                           // last commit workgroup of the launch before, into the set the next launch consumes)
    kPipeWinners = 68,     // buckets claimed in the frame = entries its commit phase will insert
    kPipeSetStride = 32,
    kPipeScanB = 64 + 3 * 32,   // list B of the set (a second line per set, kPipeSetStride apart like the first)
    kPipeCommitDone = 64 + 6 * 32,   // overflow list, pipelined: tag of the frame whose commit phase has finished (a line of its own: every
                                     // claim / walk workgroup of the launch polls it.  64 copies on lines of their own, workgroup b
                                     // polling copy b % 64, changed nothing: 184.6 against 184.6 us on C2 -- it is not the polling)
    kNumCounters = 64 + 7 * 32
};

// Everything a kernel needs about the frame, passed by value in the kernel
// argument segment (the reference uploads it to __constant__ memory twice per
// frame, SDF_Hashtable.cpp:21,33).
struct FrameParams {
    float T[16];          // global_transform  (camera -> world), row-major
    float Tinv[16];       // inv_global_transform
    float proj[9];        // "kinectProjectionMatrix", VoxelUtils.cu:24
    float voxelSize;
    float truncation;
    float weightMax;
    int32_t width, height;
    int32_t semantics;
    uint32_t numBuckets;  // logical table size (hash modulus)
    uint32_t bucketSize;
    uint32_t bucketLo;    // this context owns buckets [bucketLo, bucketHi)
    uint32_t bucketHi;
    uint32_t numVoxelBlocks;
    uint32_t epoch;       // bucket-lock epoch of this frame, 1..kMaxClaimEpoch (the host clears the claim words on wrap)
    float allocBand;      // 0: a pixel demands its surface block only (reference); > 0: +- band along the ray
    uint32_t flags;       // kFlag*: opt-in extensions (all 0 = the live reference path)
    uint32_t listSize;    // attachedLinkedListSize: iterations of the chain loop (VoxelUtils.cu:391-392)
    float truncScale;     // getTruncation, :261-264 (kFlagDepthTruncation)
    uint32_t weightSample;   // integrationWeightSample, :827 (kFlagWeightSample)
};

constexpr uint32_t kFlagOverflow = 1u;          // overflow linked list (dead code in the reference, :384-411, :458-539, :578-602)
constexpr uint32_t kFlagBandDda = 2u;           // band allocation by a block DDA along the normal (:632-703) instead of ray samples
constexpr uint32_t kFlagDepthTruncation = 4u;   // truncation + truncScale * depth (:815)
constexpr uint32_t kFlagWalkShort = 32u;        // the table walk takes 4 entries per lane instead of 8 (option "walk_entries")
constexpr uint32_t kFlagWalkNt = 16u;           // the table walk's ptr loads are non-temporal (option "walk_nt")
constexpr uint32_t kFlagBandRayDda = 64u;       // band allocation by the block DDA along the viewing ray (VH_BAND_RAY_DDA)
constexpr uint32_t kFlagDebugNoProbe = 1u << 30;   // diagnostics builds only (VH_DEBUG_SKIP_ROLES): claim tiles return before probing
constexpr uint32_t kFlagWeightSample = 8u;      // weight = max(integrationWeightSample * 1.5 * (1 - depth01), 1) (:808-811, :827)
constexpr int kLookAhead = 10;                  // free-slot search behind a full bucket: j < 10 (:475-478)

struct DevPtrs {
    uint32_t *heap;
    VoxelEntry *table;        // (bucketHi-bucketLo)*bucketSize entries
    VoxelEntry *compact;
    unsigned long long *claim;  // one epoch-stamped claim word per owned bucket
    Voxel *blocks;
    int32_t *counters;        // Counter[]
    int4 *candidates;         // {x,y,z,rank} of this frame's contenders
    uint32_t candCapacity;
    uint32_t *candTarget;     // overflow list: entry index of the free slot a contender found outside its full home bucket (~0u: none)
    uint32_t *gcMarks;        // one bit per entry: due for deletion by the running vh_delete_blocks / vh_garbage_collect
    uint32_t *compactMask;    // multi-camera frames: cameras that see compact entry i
    uint32_t *bucketBits;     // one bit per owned bucket: holds at least one entry
    uint32_t *macroBits;      // raycast: one bit per hashed 4x4x4-block macro cell that holds a block
};

// camera packet of the sharded path: 16 floats pose, 16 floats inverse, W*H camera-z plane
constexpr int kPacketHeader = 32;
// sensor-depth packet (VH_PACKET_U16): the same 32 floats, then K_inv row 2 (3 floats) and the depth
// unit (5000 = 1 m, CameraTrackingUtils.cu:64), then the W*H uint16 depth image itself: half the
// bytes of the float plane on the wire, and the owner recomputes the camera z with preProcess's
// arithmetic, bit for bit
constexpr int kPacketHeaderU16 = 36;

struct int3_ { int x, y, z; };

// float -> int exactly as CUDA's cvt.rzi.s32.f32 (what int(f), make_int2(float,
// float) and __float2int_rz compile to in the reference): truncate, saturate,
// NaN -> 0.  v_cvt_i32_f32 has the same contract on gfx950.
__device__ __forceinline__ int f2i_rz(float x)
{
    int r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

// calculateHash, VoxelUtils.cu:250-259 (unsigned modulo)
__device__ __forceinline__ uint32_t hash_block(int x, int y, int z, uint32_t numBuckets)
{
    const uint32_t h = ((uint32_t)x * 73856093u) ^ ((uint32_t)y * 19349669u) ^ ((uint32_t)z * 83492791u);
    // a 32-bit remainder by a run-time value is ~25 instructions on CDNA; every BASELINE
    // config uses a power-of-two bucket count, where it is one AND (wave-uniform branch)
    if ((numBuckets & (numBuckets - 1u)) == 0u) return h & (numBuckets - 1u);
    return h % numBuckets;
}

// Division by a fixed divisor.  The compiler's IEEE fp32 division is, when none of v_div_scale's rescaling
// cases applies, exactly: r0 = rcp(d); r1 = fma(fma(-d, r0, 1), r0, r0); q0 = n * r1; q1 = fma(fma(-d, q0, n),
// r1, q0); q = fma(fma(-d, q1, n), r1, q1).  r1 depends on the divisor alone (a pixel's ray direction over all
// the cubes it is tested against), so it is computed once and a division costs 5 instructions instead of 13, with the same bits.  The rescaling cases (denormal or huge
// operands or quotients, a tiny numerator) are kept out by range checks at the call sites: 2^-40 <= |d| <=
// 2^40 and 2^-50 <= |n| <= 2^50; anything else takes the plain division.
__device__ __forceinline__ bool fast_range(float x, float lo, float hi)
{
    const float a = __builtin_fabsf(x);
    return a >= lo && a <= hi;
}
__device__ __forceinline__ float refined_rcp(float d)
{
    const float r0 = __builtin_amdgcn_rcpf(d);
    return __builtin_fmaf(__builtin_fmaf(-d, r0, 1.0f), r0, r0);
}
__device__ __forceinline__ float div_fixed(float n, float d, float r1)
{
    const float q0 = n * r1;
    const float q1 = __builtin_fmaf(__builtin_fmaf(-d, q0, n), r1, q0);
    return __builtin_fmaf(__builtin_fmaf(-d, q1, n), r1, q1);
}

// world2Voxel, VoxelUtils.cu:280-287: true divide, round half away from zero
__device__ __forceinline__ int world2voxel1(float p, float voxelSize)
{
    const float q = p / voxelSize;
    return f2i_rz(q + __builtin_copysignf(0.5f, q));
}

// voxel2Block, VoxelUtils.cu:266-278: floor division by 8
__device__ __forceinline__ int voxel2block1(int v)
{
    if (v < 0) v = (int)((uint32_t)v - 7u);
    return v / 8;
}

// world2Block with the division by voxelSize as div_fixed (r1 = refined_rcp(voxelSize), ok = voxelSize within its
// range): the same bits, 5 instructions per division instead of 13; a coordinate outside the numerator's range
// (0 included) takes the plain division
struct FixedDivisor {
    float d, r1;
    bool ok;
    __device__ __forceinline__ explicit FixedDivisor(float divisor)
        : d(divisor), r1(refined_rcp(divisor)), ok(fast_range(divisor, 0x1p-40f, 0x1p40f)) {}
    __device__ __forceinline__ FixedDivisor(float divisor, float refined, bool inRange) : d(divisor), r1(refined), ok(inRange) {}
    __device__ __forceinline__ float divide(float n) const
    {
        return (ok && fast_range(n, 0x1p-50f, 0x1p50f)) ? div_fixed(n, d, r1) : n / d;
    }
};

__device__ __forceinline__ int world2voxel1(float p, const FixedDivisor &vs)
{
    const float q = vs.divide(p);
    return f2i_rz(q + __builtin_copysignf(0.5f, q));
}

__device__ __forceinline__ int3_ world2block(float x, float y, float z, const FixedDivisor &vs)
{
    int3_ b;
    b.x = voxel2block1(world2voxel1(x, vs));
    b.y = voxel2block1(world2voxel1(y, vs));
    b.z = voxel2block1(world2voxel1(z, vs));
    return b;
}

__device__ __forceinline__ int3_ world2block(float x, float y, float z, float voxelSize)
{
    int3_ b;
    b.x = voxel2block1(world2voxel1(x, voxelSize));
    b.y = voxel2block1(world2voxel1(y, voxelSize));
    b.z = voxel2block1(world2voxel1(z, voxelSize));
    return b;
}

// float4x4::operator*(float4), cuda_SimpleMatrixUtil.h:888-896, rows summed
// left to right
__device__ __forceinline__ float4 mat4_mul(const float *m, float x, float y, float z, float w)
{
    float4 r;
    r.x = m[0] * x + m[1] * y + m[2] * z + m[3] * w;
    r.y = m[4] * x + m[5] * y + m[6] * z + m[7] * w;
    r.z = m[8] * x + m[9] * y + m[10] * z + m[11] * w;
    r.w = m[12] * x + m[13] * y + m[14] * z + m[15] * w;
    return r;
}

// project, VoxelUtils.cu:770-777: M*p, true divides by .z, implicit float->int
// (The two divisions share their divisor, and div_fixed with one refined reciprocal gives the same bits in 13 instead of 26
// instructions where its range conditions hold -- measured in round 4, same box (profiles/r04_ab_frame_project.txt): the
// range tests and the branch to the plain division cost more than the divisions saved: C2 launch 17.65 -> 18.20 us, C3
// 67.7 -> 68.5, the TSDF-update launch of C3 11.6 -> 12.4.  Not adopted.)
__device__ __forceinline__ void project(const float *m, float x, float y, float z, int &sx, int &sy)
{
    const float qx = m[0] * x + m[1] * y + m[2] * z;
    const float qy = m[3] * x + m[4] * y + m[5] * z;
    const float qz = m[6] * x + m[7] * y + m[8] * z;
    sx = f2i_rz(qx / qz);
    sy = f2i_rz(qy / qz);
}

// blockInFrustum, VoxelUtils.cu:344-359 (REFERENCE) / corrected variant (PINHOLE)
// T / Tinv: the camera's pose and inverse (fp.T / fp.Tinv, or a camera packet's)
__device__ __forceinline__ bool block_in_frustum(const FrameParams &fp, const float *T, const float *Tinv, int bx,
                                                 int by, int bz)
{
    const float wx = (float)(int)((uint32_t)bx * 8u) * fp.voxelSize;   // block2World :289-304
    const float wy = (float)(int)((uint32_t)by * 8u) * fp.voxelSize;
    const float wz = (float)(int)((uint32_t)bz * 8u) * fp.voxelSize;
    float4 c;
    if (fp.semantics == VH_SEM_REFERENCE) {
        c = mat4_mul(T, wx, wy, wz, 1.0f);
    } else {
        c = mat4_mul(Tinv, wx, wy, wz, 1.0f);
        if (!(c.z > 0.0f)) return false;
    }
    int sx, sy;
    project(fp.proj, c.x, c.y, c.z, sx, sy);
    return sx < fp.width && sx >= 0 && sy < fp.height && sy >= 0;
}

__device__ __forceinline__ bool block_in_frustum(const FrameParams &fp, int bx, int by, int bz)
{
    return block_in_frustum(fp, fp.T, fp.Tinv, bx, by, bz);
}

// Contender ranks are 32 bits: camera (5) | launch rank of the pixel (21) | band sample (6).
// A pixel has at most 63 band samples (k <= 62), so the all-ones rank never occurs and a claim
// word can never equal consumed_word() of its epoch.
constexpr int kMaxBandSamples = 64;
constexpr uint32_t kRankSampleBits = 6, kRankCameraShift = 27;

// Position of pixel (x,y) in the launch order of the reference's grid of 16x16
// tiles (VoxelUtils.cu:610-611,710-712).  The lowest rank contending for a
// bucket wins it for the frame (SURVEY.md 8(c) determinism rule).
__device__ __forceinline__ uint32_t launch_rank(int x, int y, int width)
{
    const uint32_t tilesX = (uint32_t)(width + 15) >> 4;
    return ((((uint32_t)y >> 4) * tilesX + ((uint32_t)x >> 4)) << 8) + (((uint32_t)y & 15u) << 4) + ((uint32_t)x & 15u);
}

// Claim word of a contender, 64 bits: [epoch:10 | 0xfffffffe - rank:32 | f:3 | slot:19].
// Newer epochs beat stale words and, within an epoch, the lowest rank gives the largest word, so
// atomicMax over the contenders of a bucket leaves the winner's word: the thread a sequential run of
// the reference grid would have let through the atomicExch (VoxelUtils.cu:444-445).  The word also
// says WHO won (slot = index of the contender's record in the candidate list, hence its key) and
// WHERE the entry goes (f = first free slot of the bucket as the contender saw it): the pipelined
// frame reads both while the insertion itself is still in flight (vh_frame.hip).  Once the winner has
// been served the word is replaced by the epoch's largest value, so the bucket stays locked for the
// rest of the epoch even if allocBlocks runs again before the next reset (the reference's mutex is
// never released within a frame).
// Layout: epoch 9 bits | 0xfffffffe - rank 32 bits | f 4 bits (buckets of up to 16 slots) | slot 19 bits.
#ifndef VH_CLAIM_F_BITS
#define VH_CLAIM_F_BITS 4
#endif
constexpr uint32_t kClaimSlotBits = 19, kClaimFBits = VH_CLAIM_F_BITS, kClaimRankShift = kClaimSlotBits + kClaimFBits, kClaimEpochShift = kClaimRankShift + 32;
constexpr uint32_t kMaxClaimEpoch = (1u << (64 - kClaimEpochShift)) - 1u;      // 511: after that many epochs the words are cleared
constexpr uint32_t kMaxCandidates = (1u << kClaimSlotBits) - 1u;      // candidate records per lock epoch
constexpr uint32_t kMaxPipelinedBucket = 1u << kClaimFBits;           // f must name every slot of the bucket

__device__ __forceinline__ unsigned long long claim_word(uint32_t epoch, uint32_t rank, uint32_t f, uint32_t slot)
{
    return ((unsigned long long)epoch << kClaimEpochShift) | ((unsigned long long)(0xfffffffeu - rank) << kClaimRankShift) |
           ((unsigned long long)(f & (kMaxPipelinedBucket - 1u)) << kClaimSlotBits) | (unsigned long long)slot;
}

__device__ __forceinline__ unsigned long long consumed_word(uint32_t epoch)
{
    return ((unsigned long long)epoch << kClaimEpochShift) | ((1ull << kClaimEpochShift) - 1ull);
}

__device__ __forceinline__ uint32_t claim_epoch(unsigned long long w) { return (uint32_t)(w >> kClaimEpochShift); }
__device__ __forceinline__ uint32_t claim_slot(unsigned long long w) { return (uint32_t)w & kMaxCandidates; }
__device__ __forceinline__ uint32_t claim_f(unsigned long long w) { return (uint32_t)(w >> kClaimSlotBits) & (kMaxPipelinedBucket - 1u); }

// ---- overflow list (kFlagOverflow) -------------------------------------------------------------
// The entries of one home bucket that did not fit its slots form a chain that starts in the bucket's
// LAST slot and is linked through `offset`, measured from that slot, modulo this table's entries
// (VoxelUtils.cu:388-399; a shard's chains wrap inside the shard).  offset 0 ends the chain.
__device__ __forceinline__ uint32_t chain_slot(uint32_t last, int32_t offset, uint32_t numEntries)
{
    uint32_t s = last + (uint32_t)offset;            // offsets are 1..kLookAhead-1
    if (s >= numEntries) s -= numEntries;
    return s;
}

__device__ __forceinline__ uint32_t owned_entries(const FrameParams &fp)
{
    return (fp.bucketHi - fp.bucketLo) * fp.bucketSize;
}

__device__ __forceinline__ bool entry_is(const VoxelEntry &e, int kx, int ky, int kz)
{
    return e.pos[0] == kx && e.pos[1] == ky && e.pos[2] == kz && e.ptr != VH_FREE_BLOCK;
}

// getVoxelEntry4Block with the list (:362-411): entry index of the key or ~0u; prev = its chain
// predecessor when it was found behind the bucket's last slot (~0u otherwise)
__device__ __forceinline__ uint32_t find_entry_overflow(const FrameParams &fp, const VoxelEntry *__restrict__ table,
                                                        uint32_t numEntries, uint32_t local, int kx, int ky, int kz,
                                                        uint32_t &prev)
{
    const uint32_t start = local * fp.bucketSize, last = start + fp.bucketSize - 1u;
    prev = ~0u;
    for (uint32_t i = 0; i < fp.bucketSize; ++i)
        if (entry_is(table[start + i], kx, ky, kz)) return start + i;                 // :374-381
    uint32_t i = last, before = last;
    for (uint32_t iter = 0; iter < fp.listSize; ++iter) {                             // :391-392
        const VoxelEntry curr = table[i];
        if (entry_is(curr, kx, ky, kz)) { if (i != last) prev = before; return i; }
        if (curr.offset == 0) break;                                                  // :396
        before = i;
        i = chain_slot(last, curr.offset, numEntries);                                // :398-399
    }
    return ~0u;
}

}  // namespace vh
