// vh_kernels.hip -- hand-written gfx950 kernels of the voxel-hashing TSDF path.
//
//   alloc_claim_kernel    per-pixel block key + wave-level run dedup + bucket probe;
//                         contenders stake an epoch-stamped claim on their bucket
//                         (allocBlocksKernel + the locking half of insertVoxelEntry,
//                         VoxelUtils.cu:606-705, 418-456)
//   alloc_commit_kernel   the contender that holds the claim writes the entry and
//                         pops the heap (VoxelUtils.cu:447-453, 328-334)
//   flatten_kernel        one coalesced walk over the VoxelEntry array, wave-ballot
//                         compaction of allocated in-frustum entries
//                         (flattenKernel, VoxelUtils.cu:719-749)
//   integrate_kernel      one 8^3 block per workgroup pass, 16-byte-per-lane voxel
//                         read-modify-write (integrateDepthMapKernel, VoxelUtils.cu:790-842)
//   raycast_kernel        per-pixel march through the hash (stand-in for
//                         SDFRenderer::render, SDFRenderer.cpp:210-255)
//
// All of it is integer/fp32 scalar work bound by HBM traffic and latency; there
// is no contraction to hand to MFMA.
#include "vh_device.h"

namespace vh {

// ---------------------------------------------------------------------------
// bucket probe shared by the claim kernels
// ---------------------------------------------------------------------------
// Reads the bucket of `key` the way insertVoxelEntry scans it (VoxelUtils.cu:436-456):
// present -> nothing to do; otherwise, if a free slot exists, stake a claim.
// Allocated entries always form a prefix of the bucket (insertions take the
// first free slot, nothing is ever deleted), so "present anywhere" equals the
// reference's in-order scan.
__device__ __forceinline__ void probe_and_claim(const FrameParams &fp, const DevPtrs &dp, int kx, int ky, int kz,
                                                uint32_t h, uint32_t rank, int candCounter = kCandCount)
{
    const VoxelEntry *bucket = dp.table + (size_t)(h - fp.bucketLo) * fp.bucketSize;
    bool has_free = false;
    for (uint32_t i = 0; i < fp.bucketSize; ++i) {
        const VoxelEntry e = bucket[i];
        if (e.ptr == VH_FREE_BLOCK) {
            has_free = true;
            break;                       // prefix property: nothing allocated behind a free slot
        }
        if (e.pos[0] == kx && e.pos[1] == ky && e.pos[2] == kz) return;   // already allocated
    }
    if (!has_free) return;               // bucket full: the key is dropped (no overflow list)
    atomicMax(dp.claim + (h - fp.bucketLo), claim_word(fp.epoch, rank));
    const uint32_t slot = (uint32_t)atomicAdd(dp.counters + candCounter, 1);
    if (slot < dp.candCapacity) dp.candidates[slot] = make_int4(kx, ky, kz, (int)rank);
}

// ---------------------------------------------------------------------------
// allocBlocks, phase 1
// ---------------------------------------------------------------------------
// One lane per pixel, row-major, so the float4 vertex map is read with 16-byte
// coalesced loads (1 KiB per wave instruction).  Neighbouring pixels almost
// always fall into the same 8^3 block, so each wave collapses runs of equal
// keys to their first lane before touching the table: ~300 k pixels become a
// few thousand bucket probes.  Within an image row the launch rank grows with
// x, so the first lane of a run carries the run's lowest rank.
// Truncation-band allocation (opt-in; commented out in the reference,
// VoxelUtils.cu:632-703): with fp.allocBand = b > 0 a pixel demands the blocks of
// 2*ceil(b/step)+1 points on its viewing ray at camera depths z + (k - half)*step, step = half
// a block edge; the middle sample is the surface point itself.  b = 0: that sample only.
struct PixelVertex {
    float4 v;
    int px, py;
    bool valid;
};

__device__ __forceinline__ int band_samples(const FrameParams &fp, float &step)
{
    step = 4.0f * fp.voxelSize;
    if (!(fp.allocBand > 0.0f)) return 1;
    int half = (int)__builtin_ceilf(fp.allocBand / step);
    half = min(half, (kMaxBandSamples - 1) / 2);
    return 2 * half + 1;
}

__device__ __forceinline__ PixelVertex load_pixel(const FrameParams &fp, const float4 *__restrict__ verts, int idx,
                                                  float *__restrict__ outDepth)
{
    PixelVertex p{make_float4(0.f, 0.f, 0.f, 0.f), 0, 0, false};
    if (idx < fp.width * fp.height) {
        p.v = verts[idx];
        if (outDepth) outDepth[idx] = p.v.z;                             // camera-z plane of a camera packet
        p.py = idx / fp.width;
        p.px = idx - p.py * fp.width;
        p.valid = p.v.z != 0.0f;                                         // VoxelUtils.cu:621
    }
    return p;
}

struct SampleKey {
    int kx, ky, kz;
    bool leader;       // this lane must probe / emit the key (first of a run of equal in-frustum keys)
};

// Key of band sample k of this lane's pixel, de-duplicated against the lane's own previous
// sample and against the previous lane's sample k (runs of equal keys along an image row
// collapse to their first lane; within a row the launch rank grows with x and, within a pixel,
// with k, so whoever survives carries the lowest rank of its run).
__device__ __forceinline__ SampleKey sample_key(const FrameParams &fp, const PixelVertex &p, int k, int nS, float step,
                                                int &ownX, int &ownY, int &ownZ, bool &ownHave)
{
    SampleKey r{0, 0, 0, false};
    bool want = false;
    if (p.valid) {
        const int half = (nS - 1) / 2;
        const float s = p.v.z + ((float)k - (float)half) * step;
        if (k == half || s > 0.0f) {                                     // the surface sample is never filtered (:621 only tests z != 0)
            float x = p.v.x, y = p.v.y, z = p.v.z;                       // k == half: the vertex itself, bit for bit
            if (k != half) {                                             // wave-uniform; no divide on the reference path
                const float scale = s / p.v.z;
                x = p.v.x * scale; y = p.v.y * scale; z = s;
            }
            const float4 g = mat4_mul(fp.T, x, y, z, p.v.w);             // :622, w as stored
            const int3_ b = world2block(g.x, g.y, g.z, fp.voxelSize);    // :636
            r.kx = b.x; r.ky = b.y; r.kz = b.z;
            want = block_in_frustum(fp, r.kx, r.ky, r.kz);               // :673
        }
    }
    const bool dupOwn = want && ownHave && ownX == r.kx && ownY == r.ky && ownZ == r.kz;
    if (want) { ownX = r.kx; ownY = r.ky; ownZ = r.kz; ownHave = true; }
    const int lane = threadIdx.x & (kWave - 1);
    const int pkx = __shfl_up(r.kx, 1), pky = __shfl_up(r.ky, 1), pkz = __shfl_up(r.kz, 1);
    const int ppy = __shfl_up(p.py, 1);
    const int pwant = __shfl_up((int)want, 1);
    r.leader = want && !dupOwn &&
               (lane == 0 || !pwant || ppy != p.py || pkx != r.kx || pky != r.ky || pkz != r.kz);
    return r;
}

__device__ __forceinline__ uint32_t sample_rank(const FrameParams &fp, const PixelVertex &p, int k)
{
    return (launch_rank(p.px, p.py, fp.width) << kRankSampleBits) | (uint32_t)k;
}

// the claim phase for one lane = one pixel (shared by alloc_claim_kernel and the fused frame)
__device__ __forceinline__ void claim_pixel(const FrameParams &fp, const DevPtrs &dp,
                                            const float4 *__restrict__ verts, int idx, int candCounter)
{
    const PixelVertex p = load_pixel(fp, verts, idx, nullptr);
    float step;
    const int nS = band_samples(fp, step);
    int ox = 0, oy = 0, oz = 0;
    bool oh = false;
    for (int k = 0; k < nS; ++k) {
        const SampleKey s = sample_key(fp, p, k, nS, step, ox, oy, oz, oh);
        if (!s.leader) continue;
        const uint32_t h = hash_block(s.kx, s.ky, s.kz, fp.numBuckets);
        if (h < fp.bucketLo || h >= fp.bucketHi) continue;              // not this shard's bucket
        probe_and_claim(fp, dp, s.kx, s.ky, s.kz, h, sample_rank(fp, p, k), candCounter);
    }
}

__global__ __launch_bounds__(256) void alloc_claim_kernel(const FrameParams fp, const DevPtrs dp,
                                                          const float4 *__restrict__ verts)
{
    claim_pixel(fp, dp, verts, blockIdx.x * 256 + threadIdx.x, kCandCount);
}

// Key generation for the multi-GPU exchange: the same per-pixel
// work, but the surviving keys are binned by owning shard instead of probed.  Slots in
// a bin come from one global counter per bin; to keep that word off the critical path
// (one address sustains only ~90 returning atomics per microsecond) a 1024-lane
// workgroup first counts its keys per owner in LDS and then takes one global
// atomicAdd per owner it actually has keys for.
constexpr int kGenThreads = 1024;

__global__ __launch_bounds__(kGenThreads) void generate_keys_kernel(const FrameParams fp,
                                                                    const float4 *__restrict__ verts,
                                                                    int32_t numShards, int4 *__restrict__ outBins,
                                                                    int32_t outCapacity, int32_t outBinStride,
                                                                    float *__restrict__ outDepth, uint32_t rankBase)
{
    __shared__ int ldsCount[VH_MAX_CAMERAS];
    __shared__ int ldsBase[VH_MAX_CAMERAS];
    if (outDepth && blockIdx.x == 0 && threadIdx.x < kPacketHeader)      // packet header: pose, inverse
        outDepth[(int)threadIdx.x - kPacketHeader] = threadIdx.x < 16 ? fp.T[threadIdx.x] : fp.Tinv[threadIdx.x - 16];
    const PixelVertex p = load_pixel(fp, verts, blockIdx.x * kGenThreads + threadIdx.x, outDepth);
    float step;
    const int nS = band_samples(fp, step);
    const uint32_t perShard = (fp.numBuckets + (uint32_t)numShards - 1u) / (uint32_t)numShards;
    int ox = 0, oy = 0, oz = 0;
    bool oh = false;
    for (int k = 0; k < nS; ++k) {
        if (threadIdx.x < VH_MAX_CAMERAS) ldsCount[threadIdx.x] = 0;
        __syncthreads();
        const SampleKey s = sample_key(fp, p, k, nS, step, ox, oy, oz, oh);
        uint32_t owner = 0;
        int local = 0;
        if (s.leader) {
            owner = hash_block(s.kx, s.ky, s.kz, fp.numBuckets) / perShard;
            local = atomicAdd(&ldsCount[owner], 1);
        }
        __syncthreads();
        if ((int)threadIdx.x < numShards && ldsCount[threadIdx.x] > 0)
            ldsBase[threadIdx.x] = atomicAdd(&outBins[(size_t)threadIdx.x * outBinStride].x, ldsCount[threadIdx.x]);
        __syncthreads();
        if (s.leader) {
            int4 *bin = outBins + (size_t)owner * outBinStride;           // record 0 = {count,0,0,0}
            const int slot = ldsBase[owner] + local + 1;
            if (slot < outCapacity) bin[slot] = make_int4(s.kx, s.ky, s.kz, (int)(rankBase + sample_rank(fp, p, k)));
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// allocBlocks, phase 2
// ---------------------------------------------------------------------------
// Exactly one contender per bucket finds its own word in the claim array: the
// one with the lowest launch rank, i.e. the thread a sequential run of the
// reference grid would have let through the atomicExch (VoxelUtils.cu:444-445).
// It takes the first free slot and pops the heap (top-down, :328-334).  An empty
// heap refuses the insertion instead of reading heap[-1].
// Raycast accelerator: "macro cells" of 4x4x4 blocks, one bit per hashed macro coordinate
// (collisions only make the ray skip less).  Set when a block inside the cell is inserted.
constexpr uint32_t kMacroBits = 1u << 20;      // 128 KB bitmap

__device__ __forceinline__ uint32_t macro_hash(int mx, int my, int mz)
{
    return (((uint32_t)mx * 73856093u) ^ ((uint32_t)my * 19349669u) ^ ((uint32_t)mz * 83492791u)) & (kMacroBits - 1u);
}

// Returns true (and the new entry) if candidate k held its bucket's claim and was inserted.
__device__ __forceinline__ bool commit_candidate(const FrameParams &fp, const DevPtrs &dp, const int4 k,
                                                 VoxelEntry &e)
{
    const uint32_t h = hash_block(k.x, k.y, k.z, fp.numBuckets);
    const uint32_t local = h - fp.bucketLo;
    if (dp.claim[local] != claim_word(fp.epoch, (uint32_t)k.w)) return false;   // lost the bucket this frame
    dp.claim[local] = consumed_word(fp.epoch);                                   // locked until the next epoch
    VoxelEntry *bucket = dp.table + (size_t)local * fp.bucketSize;
    for (uint32_t s = 0; s < fp.bucketSize; ++s) {
        if (bucket[s].ptr != VH_FREE_BLOCK) continue;
        const int addr = atomicSub(dp.counters + kHeapCounter, 1);
        if (addr < 0) {                                   // heap empty: undo, refuse
            atomicAdd(dp.counters + kHeapCounter, 1);
            atomicAdd(dp.counters + kHeapExhausted, 1);
            return false;
        }
        e.pos[0] = k.x; e.pos[1] = k.y; e.pos[2] = k.z;
        e.ptr = (int)(dp.heap[addr] * (uint32_t)kBlockVoxels);
        e.offset = 0;
        bucket[s] = e;
        atomicOr(dp.bucketBits + (local >> 5), 1u << (local & 31u));
        const uint32_t hm = macro_hash(k.x >> 2, k.y >> 2, k.z >> 2);
        atomicOr(dp.macroBits + (hm >> 5), 1u << (hm & 31u));
        atomicAdd(dp.counters + kAllocatedTotal, 1);
        return true;
    }
    return false;
}

__global__ __launch_bounds__(256) void alloc_commit_kernel(const FrameParams fp, const DevPtrs dp)
{
    int n = dp.counters[kCandCount];
    if ((uint32_t)n > dp.candCapacity) n = (int)dp.candCapacity;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        VoxelEntry e;
        (void)commit_candidate(fp, dp, dp.candidates[i], e);
    }
    // the last workgroup to finish re-arms the per-frame counters
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const int ticket = atomicAdd(dp.counters + kCommitTicket, 1);
        if (ticket == (int)gridDim.x - 1) {
            dp.counters[kLastCandidates] = dp.counters[kCandCount];
            dp.counters[kCandCount] = 0;
            dp.counters[kCompactCount] = 0;
            dp.counters[kCommitTicket] = 0;
        }
    }
}

// ---------------------------------------------------------------------------
// flattenIntoBuffer
// ---------------------------------------------------------------------------
// Only `ptr` decides whether an entry is live, and all but a few thousand of
// the millions of entries are free.  Two ways to stream the ptr dwords:
//   kWalkStrided  every lane reads just the ptr dword of its entries (stride 20 B: a
//                 wave instruction covers 1280 contiguous bytes, every fetched line
//                 is consumed across the loads in flight);
//   kWalkWide     every lane reads 16-byte chunks, a wave instruction 1 KiB, the
//                 best-coalesced shape there is.  20-byte records repeat every 5
//                 chunks (80 B = 4 entries), so chunk c holds the ptr of entry
//                 (4c + d - 3) / 5 in dword d = 3,-,0,1,2 for c mod 5 = 0..4 and no
//                 staging through LDS is needed to find it.
// The rare live entries are re-read in full and frustum-tested; slots in the compact
// list are taken with one atomic per wave (wave scan of the per-lane hit counts).  The
// reference also clears the whole compact table first (VoxelUtils.cu:757-758, its own
// TODO calls it redundant); that pass is dropped.
constexpr int kFlattenThreads = 256;
constexpr int kEntriesPerLane = 8;
constexpr int kChunksPerLane = 8;
enum WalkKind : int {
    kWalkStridedNT = 0, kWalkStrided = 1, kWalkWide = 2, kWalkStridedBallot = 3, kWalkIndexed = 4, kWalkPersistent = 5,
    kWalkMask = 6      // fused frame only: launch 1 stores allocation masks, launch 2 consumes them
};

// First compact slot for this lane's `myCount` hits (one atomicAdd per wave that has any).
__device__ __forceinline__ int reserve_compact_slots(const DevPtrs &dp, int counter, int myCount)
{
    if (__ballot(myCount != 0) == 0ull) return -1;
    const int lane = threadIdx.x & (kWave - 1);
    int incl = myCount;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const int n = __shfl_up(incl, d);
        if (lane >= d) incl += n;
    }
    int base = 0;
    if (lane == kWave - 1) base = atomicAdd(dp.counters + counter, incl);
    base = __shfl(base, kWave - 1);
    return base + incl - myCount;
}

__device__ __forceinline__ bool entry_visible(const FrameParams &fp, const DevPtrs &dp, uint32_t e)
{
    const VoxelEntry ent = dp.table[e];
    return block_in_frustum(fp, ent.pos[0], ent.pos[1], ent.pos[2]);        // VoxelUtils.cu:732
}

// strided walk with one ballot + atomic per unrolled entry slot (hits are rare on small scenes)
__device__ __forceinline__ void walk_load_tile(const DevPtrs &dp, uint32_t numEntries, uint32_t tileIndex,
                                               int32_t (&ptrs)[kEntriesPerLane])
{
    const uint32_t tile = tileIndex * (kFlattenThreads * kEntriesPerLane);
    const int32_t *words = reinterpret_cast<const int32_t *>(dp.table);
#pragma unroll
    for (int j = 0; j < kEntriesPerLane; ++j) {
        const uint32_t e = tile + j * kFlattenThreads + threadIdx.x;
        ptrs[j] = (e < numEntries) ? words[(size_t)e * kEntryDwords + 3] : VH_FREE_BLOCK;
    }
}

__device__ __forceinline__ void walk_process_tile(const FrameParams &fp, const DevPtrs &dp, uint32_t tileIndex,
                                                  const int32_t (&ptrs)[kEntriesPerLane], int counter)
{
    const uint32_t tile = tileIndex * (kFlattenThreads * kEntriesPerLane);
    bool any = false;
#pragma unroll
    for (int j = 0; j < kEntriesPerLane; ++j) any |= (ptrs[j] != VH_FREE_BLOCK);
    if (__ballot(any) == 0ull) return;
    const int lane = threadIdx.x & (kWave - 1);
#pragma unroll
    for (int j = 0; j < kEntriesPerLane; ++j) {
        bool hit = false;
        VoxelEntry ent;
        if (ptrs[j] != VH_FREE_BLOCK) {
            ent = dp.table[tile + j * kFlattenThreads + threadIdx.x];
            hit = block_in_frustum(fp, ent.pos[0], ent.pos[1], ent.pos[2]);   // VoxelUtils.cu:732
        }
        const unsigned long long mask = __ballot(hit);
        if (mask == 0ull) continue;
        int base = 0;
        const int leaderLane = __ffsll((long long)mask) - 1;
        if (lane == leaderLane) base = atomicAdd(dp.counters + counter, __popcll(mask));
        base = __shfl(base, leaderLane);
        if (hit) dp.compact[base + __popcll(mask & ((1ull << lane) - 1ull))] = ent;
    }
}

__device__ __forceinline__ void flatten_tile_ballot(const FrameParams &fp, const DevPtrs &dp, uint32_t numEntries,
                                                    uint32_t tileIndex, int counter)
{
    int32_t ptrs[kEntriesPerLane];
    walk_load_tile(dp, numEntries, tileIndex, ptrs);
    walk_process_tile(fp, dp, tileIndex, ptrs, counter);
}

// Persistent form of the same walk for tables far larger than the Infinity Cache: a workgroup
// strides over tiles and issues the ptr loads of its NEXT tile before it works through the live
// entries of the current one (re-read, frustum test, returning atomic, store: microseconds of
// latency during which the one-shot form has no streaming loads in flight).
__device__ __forceinline__ void flatten_tiles_persistent(const FrameParams &fp, const DevPtrs &dp,
                                                         uint32_t numEntries, uint32_t firstTile, uint32_t stride,
                                                         int counter)
{
    const uint32_t numTiles = (numEntries + kFlattenThreads * kEntriesPerLane - 1) / (kFlattenThreads * kEntriesPerLane);
    uint32_t t = firstTile;
    if (t >= numTiles) return;
    int32_t cur[kEntriesPerLane], nxt[kEntriesPerLane];
    walk_load_tile(dp, numEntries, t, cur);
    for (;;) {
        const uint32_t n = t + stride;
        const bool more = n < numTiles;
        if (more) walk_load_tile(dp, numEntries, n, nxt);
        walk_process_tile(fp, dp, t, cur, counter);
        if (!more) break;
#pragma unroll
        for (int j = 0; j < kEntriesPerLane; ++j) cur[j] = nxt[j];
        t = n;
    }
}

// NOT the reference algorithm (opt-in, "walk_index"): instead of visiting every VoxelEntry,
// walk the bucket-occupancy bitmap (1 bit per bucket, maintained by the commit phase) and read
// only the buckets that hold entries.  One lane per 32-bucket word; the compact SET is the
// same, the bytes moved are numBuckets/8 + 100 per non-empty bucket instead of 20*N.
__device__ __forceinline__ void flatten_index_tile(const FrameParams &fp, const DevPtrs &dp, uint32_t tileIndex,
                                                   int counter)
{
    const uint32_t owned = fp.bucketHi - fp.bucketLo;
    const uint32_t numWords = (owned + 31u) / 32u;
    const uint32_t w = tileIndex * kFlattenThreads + threadIdx.x;
    uint32_t bits = (w < numWords) ? dp.bucketBits[w] : 0u;
    const int lane = threadIdx.x & (kWave - 1);
    while (__ballot(bits != 0u) != 0ull) {
        const bool have = bits != 0u;
        const uint32_t bucket = w * 32u + (have ? (uint32_t)__ffs((int)bits) - 1u : 0u);
        if (have) bits &= bits - 1u;
        bool more = have;                   // entries form a prefix of the bucket
        for (uint32_t s = 0; s < fp.bucketSize; ++s) {
            VoxelEntry ent;
            bool hit = false;
            if (more) {
                ent = dp.table[(size_t)bucket * fp.bucketSize + s];
                more = ent.ptr != VH_FREE_BLOCK;
                hit = more && block_in_frustum(fp, ent.pos[0], ent.pos[1], ent.pos[2]);
            }
            const unsigned long long mask = __ballot(hit);
            if (__ballot(more) == 0ull && mask == 0ull) break;
            if (mask == 0ull) continue;
            int base = 0;
            const int leaderLane = __ffsll((long long)mask) - 1;
            if (lane == leaderLane) base = atomicAdd(dp.counters + counter, __popcll(mask));
            base = __shfl(base, leaderLane);
            if (hit) dp.compact[base + __popcll(mask & ((1ull << lane) - 1ull))] = ent;
        }
    }
}

// tileIndex: index of this workgroup among the `walkBlocks` workgroups doing the walk
template <int kKind>
__device__ __forceinline__ void flatten_tile(const FrameParams &fp, const DevPtrs &dp, uint32_t numEntries,
                                             uint32_t tileIndex, int counter, uint32_t walkBlocks)
{
    if constexpr (kKind == kWalkStridedBallot) {
        flatten_tile_ballot(fp, dp, numEntries, tileIndex, counter);
        return;
    }
    if constexpr (kKind == kWalkPersistent) {
        flatten_tiles_persistent(fp, dp, numEntries, tileIndex, walkBlocks, counter);
        return;
    }
    if constexpr (kKind == kWalkIndexed) {
        flatten_index_tile(fp, dp, tileIndex, counter);
        return;
    }
    uint32_t ent[kEntriesPerLane];          // entry index of each candidate, or ~0u
    uint32_t hits = 0;                      // bit j: entry j is live and in the frustum
    if constexpr (kKind == kWalkWide) {
        static_assert(kChunksPerLane == kEntriesPerLane, "one candidate entry per chunk");
        const uint32_t numChunks = (uint32_t)(((uint64_t)numEntries * 20u + 15u) / 16u);
        const uint32_t base = tileIndex * (kFlattenThreads * kChunksPerLane);
        const uint4 *chunks = reinterpret_cast<const uint4 *>(dp.table);
        uint4 v[kChunksPerLane];
#pragma unroll
        for (int j = 0; j < kChunksPerLane; ++j) {
            const uint32_t c = base + j * kFlattenThreads + threadIdx.x;
            v[j] = (c < numChunks) ? chunks[c] : make_uint4(~0u, ~0u, ~0u, ~0u);
        }
#pragma unroll
        for (int j = 0; j < kChunksPerLane; ++j) {
            const uint32_t c = base + j * kFlattenThreads + threadIdx.x;
            const uint32_t m = c % 5u;
            const uint32_t d = (m == 0u) ? 3u : m - 2u;                 // m == 1: no ptr in this chunk
            const uint32_t word = (d == 0u) ? v[j].x : (d == 1u) ? v[j].y : (d == 2u) ? v[j].z : v[j].w;
            const uint32_t e = (4u * c + d - 3u) / 5u;
            const bool live = (m != 1u) && (word != (uint32_t)VH_FREE_BLOCK) && (e < numEntries);
            ent[j] = live ? e : ~0u;
        }
    } else {
        const uint32_t tile = tileIndex * (kFlattenThreads * kEntriesPerLane);
        const int32_t *words = reinterpret_cast<const int32_t *>(dp.table);
        int32_t ptrs[kEntriesPerLane];
#pragma unroll
        for (int j = 0; j < kEntriesPerLane; ++j) {
            const uint32_t e = tile + j * kFlattenThreads + threadIdx.x;
            const int32_t *w = words + (size_t)e * kEntryDwords + 3;
            if (e >= numEntries) ptrs[j] = VH_FREE_BLOCK;
            else ptrs[j] = (kKind == kWalkStridedNT) ? __builtin_nontemporal_load(w) : *w;
        }
#pragma unroll
        for (int j = 0; j < kEntriesPerLane; ++j)
            ent[j] = (ptrs[j] != VH_FREE_BLOCK) ? tile + j * kFlattenThreads + threadIdx.x : ~0u;
    }
#pragma unroll
    for (int j = 0; j < kEntriesPerLane; ++j)
        if (ent[j] != ~0u && entry_visible(fp, dp, ent[j])) hits |= 1u << j;
    int slot = reserve_compact_slots(dp, counter, __popc(hits));
    if (slot < 0) return;
#pragma unroll
    for (int j = 0; j < kEntriesPerLane; ++j)
        if ((hits >> j) & 1u) dp.compact[slot++] = dp.table[ent[j]];
}

template <int kKind>
__global__ __launch_bounds__(kFlattenThreads) void flatten_kernel(const FrameParams fp, const DevPtrs dp,
                                                                  uint32_t numEntries)
{
    flatten_tile<kKind>(fp, dp, numEntries, blockIdx.x, kCompactCount, gridDim.x);
}

// ---------------------------------------------------------------------------
// integrateDepthMap
// ---------------------------------------------------------------------------
// A workgroup of 256 lanes owns one 8^3 block per pass: lane t updates voxels
// 2t and 2t+1 (neighbours in x), so the block moves as 16-byte-per-lane
// coalesced loads and stores (4 KiB in, 4 KiB out) instead of the reference's
// 8-byte accesses.  The occupied count never leaves the device: the grid is a
// fixed size and strides over the compact list.
// depth(x,y) = depthBase[stride*(y*W+x)]: stride 4 from &verts[0].z (float4 vertex map),
// stride 1 for the camera-z plane of a camera packet.
__device__ __forceinline__ bool tsdf_update(const FrameParams &fp, const float *Tinv,
                                            const float *__restrict__ depthBase, int stride, int vx, int vy, int vz,
                                            float &sdfOut, float &wOut)
{
    float cx, cy, cz;
    if (fp.semantics == VH_SEM_REFERENCE) {
        // VoxelUtils.cu:797-800: inverse pose on the voxel INDEX, truncate, then metres
        const float4 r = mat4_mul(Tinv, (float)vx, (float)vy, (float)vz, 1.0f);
        cx = (float)f2i_rz(r.x) * fp.voxelSize;
        cy = (float)f2i_rz(r.y) * fp.voxelSize;
        cz = (float)f2i_rz(r.z) * fp.voxelSize;
    } else {
        const float4 r = mat4_mul(Tinv, (float)vx * fp.voxelSize, (float)vy * fp.voxelSize,
                                  (float)vz * fp.voxelSize, 1.0f);
        cx = r.x; cy = r.y; cz = r.z;
    }
    int sx, sy;
    project(fp.proj, cx, cy, cz, sx, sy);                                        // :801
    if (sx < 0 || sx >= fp.width || sy < 0 || sy >= fp.height) return false;     // :803
    const float depth = depthBase[(size_t)stride * ((size_t)sy * fp.width + sx)];   // :805
    if (depth <= 0.0f) return false;                                             // :806
    float sdf = depth - cz;                                                      // :813
    if (!(sdf > -fp.truncation)) return false;                                   // :818
    sdf = (sdf >= 0.0f) ? __builtin_fminf(fp.truncation, sdf) : __builtin_fmaxf(-fp.truncation, sdf);
    // combineVoxel, :779-787, current sample {sdf, 0.1f} (:829)
    const float ow = wOut, os = sdfOut;
    sdfOut = ((os * ow) + (sdf * 0.1f)) / (ow + 0.1f);
    wOut = __builtin_fminf(fp.weightMax, ow + 0.1f);
    return true;
}

// the 256 lanes of a workgroup update the 8^3 block of entry e from the float4 vertex map
__device__ __forceinline__ void integrate_block(const FrameParams &fp, const DevPtrs &dp, const VoxelEntry &e,
                                                const float4 *__restrict__ verts)
{
    const int lin = 2 * (int)threadIdx.x;        // linearizeVoxelPos: z*64 + y*8 + x  (:311-317)
    const int tx = lin & 7, ty = (lin >> 3) & 7, tz = lin >> 6;
    const int bx = (int)((uint32_t)e.pos[0] * 8u) + tx;     // block2Voxel + threadIdx (:793-796)
    const int by = (int)((uint32_t)e.pos[1] * 8u) + ty;
    const int bz = (int)((uint32_t)e.pos[2] * 8u) + tz;
    float4 *cell = reinterpret_cast<float4 *>(dp.blocks + (size_t)e.ptr + lin);
    float4 v = *cell;                            // {sdf0, w0, sdf1, w1}
    const float *depthBase = reinterpret_cast<const float *>(verts) + 2;   // &verts[0].z
    const bool u0 = tsdf_update(fp, fp.Tinv, depthBase, 4, bx, by, bz, v.x, v.y);
    const bool u1 = tsdf_update(fp, fp.Tinv, depthBase, 4, bx + 1, by, bz, v.z, v.w);
    if (u0 || u1) *cell = v;
}

__global__ __launch_bounds__(256) void integrate_kernel(const FrameParams fp, const DevPtrs dp,
                                                        const float4 *__restrict__ verts)
{
    const int count = dp.counters[kCompactCount];
    for (int b = blockIdx.x; b < count; b += gridDim.x) integrate_block(fp, dp, dp.compact[b], verts);
}

// ---------------------------------------------------------------------------
// the fused frame: SDF_Hashtable::integrate in two launches
// ---------------------------------------------------------------------------
// Launch 1 runs the per-pixel claim phase and the table walk side by side: both only
// READ the hash table (claims go to the claim words, hits to the compact list), so the
// latency-bound pixel work hides under the bandwidth-bound walk.  The walk therefore
// sees the table as it was at the start of the frame; the entries this frame inserts
// are appended to the compact list by launch 2 -- they pass the frustum test by
// construction (allocBlocks tested the same key against the same pose, :673 / :732).
template <int kKind>
__global__ __launch_bounds__(256) void frame_scan_claim_kernel(const FrameParams fp, const DevPtrs dp,
                                                               const float4 *__restrict__ verts,
                                                               uint32_t numEntries, uint32_t claimBlocks,
                                                               int parity)
{
    // The two roles are interleaved over the grid in proportion (block b is a claim block when
    // floor((b+1)*claim/total) steps): workgroups are dispatched roughly in index order, and
    // with all claim blocks in front a large image would fill the chip with latency-bound
    // pixel work before the first byte of the table is streamed.
    const uint32_t total = gridDim.x;
    const uint32_t claimBefore = (uint32_t)(((uint64_t)blockIdx.x * claimBlocks) / total);
    const uint32_t claimAfter = (uint32_t)(((uint64_t)(blockIdx.x + 1u) * claimBlocks) / total);
    if (claimAfter != claimBefore) {
        claim_pixel(fp, dp, verts, claimBefore * 256 + threadIdx.x, kFusedCand + parity);
    } else {
        flatten_tile<kKind>(fp, dp, numEntries, blockIdx.x - claimBefore, kScanCount + parity, total - claimBlocks);
    }
}

// ---- the mask form of the fused frame (default) --------------------------------------------
// With tens of thousands of allocated entries the walk above stops being a pure stream: every
// wave that meets a live entry re-reads it, tests it and takes a returning atomic, holding its
// slot for microseconds with no streaming load in flight (C3: 97 us against 68 us for the same
// walk over an empty table).  So launch 1 only records WHERE the live entries are -- one 64-bit
// ballot per wave instruction, stored fire-and-forget (8 bytes per 64 entries) -- and everything
// with latency in it (re-read, frustum test, compaction, TSDF update) moves to launch 2, where it
// overlaps with the block updates.
constexpr int kMaskChunkWords = 256;                   // mask words per consumer workgroup
constexpr int kMaskChunkEntries = kMaskChunkWords * 64;

__device__ __forceinline__ void walk_mask_tile(const DevPtrs &dp, uint32_t numEntries, uint32_t tileIndex)
{
    int32_t ptrs[kEntriesPerLane];
    walk_load_tile(dp, numEntries, tileIndex, ptrs);
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    // entry = tile*2048 + j*256 + wave*64 + lane  =>  word = entry / 64 = tile*32 + j*4 + wave
#pragma unroll
    for (int j = 0; j < kEntriesPerLane; ++j) {
        const unsigned long long m = __ballot(ptrs[j] != VH_FREE_BLOCK);
        if (lane == 0) dp.allocMask[(size_t)tileIndex * 32 + j * 4 + wave] = m;
    }
}

__global__ __launch_bounds__(256) void frame_mask_claim_kernel(const FrameParams fp, const DevPtrs dp,
                                                               const float4 *__restrict__ verts,
                                                               uint32_t numEntries, uint32_t claimBlocks, int parity)
{
    const uint32_t total = gridDim.x;
    const uint32_t claimBefore = (uint32_t)(((uint64_t)blockIdx.x * claimBlocks) / total);
    const uint32_t claimAfter = (uint32_t)(((uint64_t)(blockIdx.x + 1u) * claimBlocks) / total);
    if (claimAfter != claimBefore) {
        claim_pixel(fp, dp, verts, claimBefore * 256 + threadIdx.x, kFusedCand + parity);
    } else {
        walk_mask_tile(dp, numEntries, blockIdx.x - claimBefore);
    }
}

// Launch 2 of the mask form.  Workgroups [0, commitBlocks): candidates, as below.  The others
// take one chunk of 256 mask words (16384 entries) each: every lane walks the set bits of its
// word (re-read, frustum test), visible entries are gathered in LDS, ONE atomicAdd reserves
// their compact slots, then the workgroup updates their blocks one after the other.  The
// occupied count is the slot counter of this frame's parity set (read by vh_get_counters).
__global__ __launch_bounds__(256) void frame_commit_consume_kernel(const FrameParams fp, const DevPtrs dp,
                                                                   const float4 *__restrict__ verts,
                                                                   uint32_t numEntries, uint32_t commitBlocks,
                                                                   int parity)
{
    __shared__ unsigned short vis[kMaskChunkEntries];
    __shared__ int nVis, slotBase;
    __shared__ VoxelEntry newEntry;
    __shared__ int inserted;
    if (blockIdx.x >= commitBlocks) {
        const uint32_t chunk = blockIdx.x - commitBlocks;
        const uint32_t numWords = (numEntries + 63u) / 64u;
        const uint32_t w = chunk * kMaskChunkWords + threadIdx.x;
        if (threadIdx.x == 0) nVis = 0;
        __syncthreads();
        unsigned long long m = (w < numWords) ? dp.allocMask[w] : 0ull;
        while (m != 0ull) {
            const int bit = __ffsll((long long)m) - 1;
            m &= m - 1ull;
            const uint32_t e = w * 64u + (uint32_t)bit;
            if (entry_visible(fp, dp, e)) vis[atomicAdd(&nVis, 1)] = (unsigned short)(threadIdx.x * 64 + bit);
        }
        __syncthreads();
        const int n = nVis;
        if (n == 0) return;
        if (threadIdx.x == 0) slotBase = atomicAdd(dp.counters + kScanCount + parity, n);
        __syncthreads();
        const uint32_t first = chunk * kMaskChunkEntries;
        for (int i = threadIdx.x; i < n; i += 256) dp.compact[slotBase + i] = dp.table[first + vis[i]];
        for (int i = 0; i < n; ++i) integrate_block(fp, dp, dp.table[first + vis[i]], verts);
        return;
    }
    int n = dp.counters[kFusedCand + parity];
    if ((uint32_t)n > dp.candCapacity) n = (int)dp.candCapacity;
    for (int i = blockIdx.x; i < n; i += commitBlocks) {
        if (threadIdx.x == 0) {
            VoxelEntry e;
            inserted = commit_candidate(fp, dp, dp.candidates[i], e) ? 1 : 0;
            if (inserted) {
                newEntry = e;
                dp.compact[atomicAdd(dp.counters + kScanCount + parity, 1)] = e;
            }
        }
        __syncthreads();
        if (inserted) integrate_block(fp, dp, newEntry, verts);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        __threadfence();
        const int ticket = atomicAdd(dp.counters + kCommitTicket, 1);
        if (ticket == (int)commitBlocks - 1) {
            dp.counters[kLastCandidates] = n;
            dp.counters[kScanCount + (parity ^ 1)] = 0;
            dp.counters[kNewCount + (parity ^ 1)] = 0;
            dp.counters[kFusedCand + (parity ^ 1)] = 0;
            dp.counters[kCommitTicket] = 0;
        }
    }
}

// Launch 2: the first commitBlocks workgroups serve the candidates (one candidate per
// workgroup pass: lane 0 inserts, then all 256 lanes integrate the new block and it is
// appended to the compact list); the others stride over the entries the walk found.
// Only the commit workgroups take a ticket (a word that every workgroup of a large grid
// increments costs tens of microseconds): the last of them publishes the occupied count
// and clears the counter set of the other parity for the next frame.
__global__ __launch_bounds__(256) void frame_commit_integrate_kernel(const FrameParams fp, const DevPtrs dp,
                                                                     const float4 *__restrict__ verts,
                                                                     uint32_t commitBlocks, int parity)
{
    const int scanCount = dp.counters[kScanCount + parity];
    if (blockIdx.x >= commitBlocks) {
        for (int b = blockIdx.x - commitBlocks; b < scanCount; b += gridDim.x - commitBlocks)
            integrate_block(fp, dp, dp.compact[b], verts);
        return;
    }
    __shared__ VoxelEntry newEntry;
    __shared__ int inserted;
    int n = dp.counters[kFusedCand + parity];
    if ((uint32_t)n > dp.candCapacity) n = (int)dp.candCapacity;
    for (int i = blockIdx.x; i < n; i += commitBlocks) {
        if (threadIdx.x == 0) {
            VoxelEntry e;
            inserted = commit_candidate(fp, dp, dp.candidates[i], e) ? 1 : 0;
            if (inserted) {
                newEntry = e;
                dp.compact[scanCount + atomicAdd(dp.counters + kNewCount + parity, 1)] = e;
            }
        }
        __syncthreads();
        if (inserted) integrate_block(fp, dp, newEntry, verts);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        __threadfence();
        const int ticket = atomicAdd(dp.counters + kCommitTicket, 1);
        if (ticket == (int)commitBlocks - 1) {
            dp.counters[kCompactCount] = scanCount + atomicAdd(dp.counters + kNewCount + parity, 0);
            dp.counters[kLastCandidates] = n;
            dp.counters[kScanCount + (parity ^ 1)] = 0;
            dp.counters[kNewCount + (parity ^ 1)] = 0;
            dp.counters[kFusedCand + (parity ^ 1)] = 0;
            dp.counters[kCommitTicket] = 0;
        }
    }
}

// ---------------------------------------------------------------------------
// multi-camera frame on a bucket-range shard
// ---------------------------------------------------------------------------
// phase 1 for key bins that arrived from the other ranks: bin b = bins[b*binStride..],
// record 0 = {count,0,0,0}, records 1..count = {x,y,z,rank}.  One lock epoch for all
// bins; rank = camera<<24 | launch rank, so cameras are served in order.
// (binIndex, part, parts): this workgroup handles every parts-th 256-record slice of the bin.
__device__ __forceinline__ void claim_bin_slice(const FrameParams &fp, const DevPtrs &dp,
                                                const int4 *__restrict__ bins, int32_t capacity, int32_t binStride,
                                                uint32_t binIndex, uint32_t part, uint32_t parts, int candCounter)
{
    const int4 *bin = bins + (size_t)binIndex * binStride;
    int n = bin[0].x;
    if (n > capacity - 1) {
        if (part == 0 && threadIdx.x == 0) atomicAdd(dp.counters + kBinOverflow, 1);
        n = capacity - 1;
    }
    for (int i = (int)part * 256 + (int)threadIdx.x; i < n; i += (int)parts * 256) {
        const int4 k = bin[1 + i];
        const uint32_t h = hash_block(k.x, k.y, k.z, fp.numBuckets);
        if (h < fp.bucketLo || h >= fp.bucketHi) continue;
        probe_and_claim(fp, dp, k.x, k.y, k.z, h, (uint32_t)k.w, candCounter);
    }
}

__global__ __launch_bounds__(256) void claim_bins_kernel(const FrameParams fp, const DevPtrs dp,
                                                         const int4 *__restrict__ bins, int32_t capacity,
                                                         int32_t binStride)
{
    claim_bin_slice(fp, dp, bins, capacity, binStride, blockIdx.y, blockIdx.x, gridDim.x, kCandCount);
}

// cameras (bit c) whose frustum holds the block
__device__ __forceinline__ uint32_t camera_mask(const FrameParams &fp, const int *pos, int32_t numCams,
                                                const float *__restrict__ packets, size_t packetStride)
{
    uint32_t seen = 0;
    for (int c = 0; c < numCams; ++c) {
        const float *pk = packets + packetStride * c;
        if (block_in_frustum(fp, pk, pk + 16, pos[0], pos[1], pos[2])) seen |= 1u << c;
    }
    return seen;
}

// One walk over the shard's entries for ALL cameras of the step: a live entry is
// tested against every camera's frustum and appended once, with the mask of the
// cameras that see it.
__device__ __forceinline__ void flatten_multi_tile(const FrameParams &fp, const DevPtrs &dp, uint32_t numEntries,
                                                   uint32_t tileIndex, int32_t numCams,
                                                   const float *__restrict__ packets, size_t packetStride,
                                                   int counter)
{
    const uint32_t tile = tileIndex * (kFlattenThreads * kEntriesPerLane);
    const int32_t *words = reinterpret_cast<const int32_t *>(dp.table);
    int32_t ptrs[kEntriesPerLane];
#pragma unroll
    for (int j = 0; j < kEntriesPerLane; ++j) {
        const uint32_t e = tile + j * kFlattenThreads + threadIdx.x;
        ptrs[j] = (e < numEntries) ? words[(size_t)e * kEntryDwords + 3] : VH_FREE_BLOCK;
    }
    uint32_t seen[kEntriesPerLane];         // cameras whose frustum holds entry j
    int myCount = 0;
#pragma unroll
    for (int j = 0; j < kEntriesPerLane; ++j) {
        seen[j] = 0;
        if (ptrs[j] == VH_FREE_BLOCK) continue;
        const VoxelEntry ent = dp.table[tile + j * kFlattenThreads + threadIdx.x];
        seen[j] = camera_mask(fp, ent.pos, numCams, packets, packetStride);
        myCount += seen[j] != 0u;
    }
    int slot = reserve_compact_slots(dp, counter, myCount);
    if (slot < 0) return;
#pragma unroll
    for (int j = 0; j < kEntriesPerLane; ++j) {
        if (seen[j] == 0u) continue;
        dp.compact[slot] = dp.table[tile + j * kFlattenThreads + threadIdx.x];
        dp.compactMask[slot] = seen[j];
        ++slot;
    }
}

__global__ __launch_bounds__(kFlattenThreads) void flatten_multi_kernel(const FrameParams fp, const DevPtrs dp,
                                                                        uint32_t numEntries, int32_t numCams,
                                                                        const float *__restrict__ packets,
                                                                        size_t packetStride)
{
    flatten_multi_tile(fp, dp, numEntries, blockIdx.x, numCams, packets, packetStride, kCompactCount);
}

// One 8^3 block per workgroup pass, the voxels stay in registers while the cameras
// that see the block are applied in camera order (the running average is order
// dependent): 4 KiB in, 4 KiB out per block whatever the number of cameras.
__device__ __forceinline__ void integrate_block_multi(const FrameParams &fp, const DevPtrs &dp, const VoxelEntry &e,
                                                      uint32_t seen, int32_t numCams,
                                                      const float *__restrict__ packets, size_t packetStride)
{
    const int lin = 2 * (int)threadIdx.x;
    const int tx = lin & 7, ty = (lin >> 3) & 7, tz = lin >> 6;
    const int bx = (int)((uint32_t)e.pos[0] * 8u) + tx;
    const int by = (int)((uint32_t)e.pos[1] * 8u) + ty;
    const int bz = (int)((uint32_t)e.pos[2] * 8u) + tz;
    float4 *cell = reinterpret_cast<float4 *>(dp.blocks + (size_t)e.ptr + lin);
    float4 v = *cell;
    bool dirty = false;
    for (int c = 0; c < numCams; ++c) {
        if (!((seen >> c) & 1u)) continue;
        const float *pk = packets + packetStride * c;
        dirty |= tsdf_update(fp, pk + 16, pk + kPacketHeader, 1, bx, by, bz, v.x, v.y);
        dirty |= tsdf_update(fp, pk + 16, pk + kPacketHeader, 1, bx + 1, by, bz, v.z, v.w);
    }
    if (dirty) *cell = v;
}

__global__ __launch_bounds__(256) void integrate_multi_kernel(const FrameParams fp, const DevPtrs dp,
                                                              int32_t numCams, const float *__restrict__ packets,
                                                              size_t packetStride)
{
    const int count = dp.counters[kCompactCount];
    for (int b = blockIdx.x; b < count; b += gridDim.x)
        integrate_block_multi(fp, dp, dp.compact[b], dp.compactMask[b], numCams, packets, packetStride);
}

// The multi-camera frame in two launches, built like the single-camera fused frame:
// launch 1 = {claim the received key bins || walk the shard for all cameras} (both only read
// the table), launch 2 = {commit: insert, camera mask, append, integrate the new block ||
// integrate the blocks the walk found}.
__global__ __launch_bounds__(256) void frame_multi_scan_claim_kernel(const FrameParams fp, const DevPtrs dp,
                                                                     const int4 *__restrict__ bins, int32_t capacity,
                                                                     int32_t binStride, uint32_t numBins,
                                                                     uint32_t partsPerBin, uint32_t numEntries,
                                                                     int32_t numCams,
                                                                     const float *__restrict__ packets,
                                                                     size_t packetStride, int parity)
{
    const uint32_t claimBlocks = numBins * partsPerBin, total = gridDim.x;
    const uint32_t claimBefore = (uint32_t)(((uint64_t)blockIdx.x * claimBlocks) / total);
    const uint32_t claimAfter = (uint32_t)(((uint64_t)(blockIdx.x + 1u) * claimBlocks) / total);
    if (claimAfter != claimBefore)
        claim_bin_slice(fp, dp, bins, capacity, binStride, claimBefore / partsPerBin, claimBefore % partsPerBin,
                        partsPerBin, kFusedCand + parity);
    else
        flatten_multi_tile(fp, dp, numEntries, blockIdx.x - claimBefore, numCams, packets, packetStride,
                           kScanCount + parity);
}

__global__ __launch_bounds__(256) void frame_multi_commit_integrate_kernel(const FrameParams fp, const DevPtrs dp,
                                                                           int32_t numCams,
                                                                           const float *__restrict__ packets,
                                                                           size_t packetStride,
                                                                           uint32_t commitBlocks, int parity)
{
    const int scanCount = dp.counters[kScanCount + parity];
    if (blockIdx.x >= commitBlocks) {
        for (int b = blockIdx.x - commitBlocks; b < scanCount; b += gridDim.x - commitBlocks)
            integrate_block_multi(fp, dp, dp.compact[b], dp.compactMask[b], numCams, packets, packetStride);
        return;
    }
    __shared__ VoxelEntry newEntry;
    __shared__ uint32_t newMask;
    __shared__ int inserted;
    int n = dp.counters[kFusedCand + parity];
    if ((uint32_t)n > dp.candCapacity) n = (int)dp.candCapacity;
    for (int i = blockIdx.x; i < n; i += commitBlocks) {
        if (threadIdx.x == 0) {
            VoxelEntry e;
            inserted = commit_candidate(fp, dp, dp.candidates[i], e) ? 1 : 0;
            if (inserted) {
                const uint32_t seen = camera_mask(fp, e.pos, numCams, packets, packetStride);
                newEntry = e;
                newMask = seen;
                if (seen != 0u) {       // what the walk would have appended had it seen the entry
                    const int slot = scanCount + atomicAdd(dp.counters + kNewCount + parity, 1);
                    dp.compact[slot] = e;
                    dp.compactMask[slot] = seen;
                }
            }
        }
        __syncthreads();
        if (inserted && newMask != 0u) integrate_block_multi(fp, dp, newEntry, newMask, numCams, packets, packetStride);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        __threadfence();
        const int ticket = atomicAdd(dp.counters + kCommitTicket, 1);
        if (ticket == (int)commitBlocks - 1) {
            dp.counters[kCompactCount] = scanCount + atomicAdd(dp.counters + kNewCount + parity, 0);
            dp.counters[kLastCandidates] = n;
            dp.counters[kScanCount + (parity ^ 1)] = 0;
            dp.counters[kNewCount + (parity ^ 1)] = 0;
            dp.counters[kFusedCand + (parity ^ 1)] = 0;
            dp.counters[kCommitTicket] = 0;
        }
    }
}

// Zeroes the header record of the bins of `batch` frames x numShards shards before
// generate_keys_kernel fills them (bin of shard s, frame b at bins[s*binStride + b*frameStride]).
__global__ void prepare_bins_kernel(int4 *bins, int32_t numShards, int32_t binStride, int32_t batch,
                                    int32_t frameStride)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < numShards * batch)
        bins[(size_t)(i / batch) * binStride + (size_t)(i % batch) * frameStride] = make_int4(0, 0, 0, 0);
}

// ---------------------------------------------------------------------------
// raycast
// ---------------------------------------------------------------------------
// getVoxelEntry4Block, live half (VoxelUtils.cu:362-382)
__device__ __forceinline__ int lookup_block(const FrameParams &fp, const DevPtrs &dp, int kx, int ky, int kz)
{
    const uint32_t h = hash_block(kx, ky, kz, fp.numBuckets);
    if (h < fp.bucketLo || h >= fp.bucketHi) return VH_FREE_BLOCK;
    const uint32_t local = h - fp.bucketLo;
    // One bit per bucket ("holds at least one entry", set by the commit phase): a few hundred
    // KB that stay in L2, while the table itself is >100 MB.  Nearly every block a ray crosses
    // is empty space and is answered here without touching the table.
    if (!((dp.bucketBits[local >> 5] >> (local & 31u)) & 1u)) return VH_FREE_BLOCK;
    const VoxelEntry *bucket = dp.table + (size_t)local * fp.bucketSize;
    for (uint32_t i = 0; i < fp.bucketSize; ++i) {
        const VoxelEntry e = bucket[i];
        if (e.ptr == VH_FREE_BLOCK) return VH_FREE_BLOCK;      // prefix property
        if (e.pos[0] == kx && e.pos[1] == ky && e.pos[2] == kz) return e.ptr;
    }
    return VH_FREE_BLOCK;
}

// Spec (DESIGN.md "raycast", oracle/vh_oracle.c vho_raycast): samples at camera
// depth t_i = tMin + i*voxelSize, nearest-voxel classification, first pair of
// consecutive valid samples with sdf_prev > 0 >= sdf_cur, linear interpolation.
// 16x16-pixel tiles: a wave is a 16x4 patch of neighbouring rays, which walk
// the same blocks and keep the bucket / voxel lines hot in L2.
constexpr float kSkipMargin = 0.01f;     // voxels; see the empty-block skip below
// kRayBatch (template): in-block samples whose voxels are fetched together

template <int kRayBatch, bool kFastDiv>
__global__ __launch_bounds__(256) void raycast_kernel(const FrameParams fp, const DevPtrs dp, float fx, float fy,
                                                      float cx, float cy, float tMin, int nSteps,
                                                      float *__restrict__ depthOut)
{
    const int u = blockIdx.x * 16 + (threadIdx.x & 15);
    const int v = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (u >= fp.width || v >= fp.height) return;
    const float dx = ((float)u - cx) / fx;
    const float dy = ((float)v - cy) / fy;
    const float dt = fp.voxelSize;
    const float invDt = __builtin_amdgcn_rcpf(dt) * (1.0f - 1.0e-6f);   // never over-estimates a step count
    const float rcpVoxel = 1.0f / fp.voxelSize;                          // correctly rounded (world2voxel1_fast)
    // world-space ray per unit of camera depth (only used to bound empty-block skips)
    const float dirX = fp.T[0] * dx + fp.T[1] * dy + fp.T[2];
    const float dirY = fp.T[4] * dx + fp.T[5] * dy + fp.T[6];
    const float dirZ = fp.T[8] * dx + fp.T[9] * dy + fp.T[10];
    const float rayD[3] = {dirX, dirY, dirZ}, rayO[3] = {fp.T[3], fp.T[7], fp.T[11]};
    float invD[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) invD[a] = (rayD[a] != 0.0f) ? __builtin_amdgcn_rcpf(rayD[a]) : 0.0f;
    bool prevValid = false, haveKey = false, found = false, haveMacro = false, macroEmpty = false;
    float prevSdf = 0.0f, prevT = 0.0f, hit = 0.0f;
    int ckx = 0, cky = 0, ckz = 0, cptr = VH_FREE_BLOCK;
    int cmx = 0, cmy = 0, cmz = 0;
    for (int i = 0; i < nSteps; ++i) {
        const float tt = tMin + (float)i * dt;
        const float4 pw = mat4_mul(fp.T, dx * tt, dy * tt, tt, 1.0f);
        const int vx = kFastDiv ? world2voxel1_fast(pw.x, fp.voxelSize, rcpVoxel) : world2voxel1(pw.x, fp.voxelSize);
        const int vy = kFastDiv ? world2voxel1_fast(pw.y, fp.voxelSize, rcpVoxel) : world2voxel1(pw.y, fp.voxelSize);
        const int vz = kFastDiv ? world2voxel1_fast(pw.z, fp.voxelSize, rcpVoxel) : world2voxel1(pw.z, fp.voxelSize);
        const int kx = voxel2block1(vx), ky = voxel2block1(vy), kz = voxel2block1(vz);
        if (!haveKey || kx != ckx || ky != cky || kz != ckz) {
            ckx = kx; cky = ky; ckz = kz;
            haveKey = true;
            const int mx = kx >> 2, my = ky >> 2, mz = kz >> 2;          // macro cell of 4x4x4 blocks
            if (!haveMacro || mx != cmx || my != cmy || mz != cmz) {
                cmx = mx; cmy = my; cmz = mz;
                haveMacro = true;
                const uint32_t hm = macro_hash(mx, my, mz);
                macroEmpty = !((dp.macroBits[hm >> 5] >> (hm & 31u)) & 1u);
            }
            cptr = macroEmpty ? VH_FREE_BLOCK : lookup_block(fp, dp, kx, ky, kz);
        }
        if (cptr == VH_FREE_BLOCK) {
            // Empty block: every further sample inside it is invalid too, so jump to the last
            // sample that is CERTAINLY still inside (cell shrunk by kSkipMargin voxels per side:
            // 1e-2 voxel = 2e-4 m at 2 cm voxels, against ~1e-6 m of fp32 difference between this
            // linear ray model and the sample positions above).  Skipping only such samples
            // leaves the result unchanged.
            prevValid = false;
            float tExit = 3.0e38f;
            // an empty macro cell (no block in 4x4x4) is skipped whole: 32 voxels per side
            const int cell[3] = {macroEmpty ? cmx * 32 : kx * 8, macroEmpty ? cmy * 32 : ky * 8,
                                 macroEmpty ? cmz * 32 : kz * 8};
            const float span = macroEmpty ? 31.5f : 7.5f;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                // the cell spans voxel centres c .. c+span-0.5, i.e. world [(c-0.5)vs, (c+span)vs)
                const float lo = ((float)cell[a] - 0.5f + kSkipMargin) * fp.voxelSize;
                const float hi = ((float)cell[a] + span - kSkipMargin) * fp.voxelSize;
                // approximate reciprocals (1 ulp) are fine here: the margin absorbs them
                if (rayD[a] > 0.0f) tExit = __builtin_fminf(tExit, (hi - rayO[a]) * invD[a]);
                else if (rayD[a] < 0.0f) tExit = __builtin_fminf(tExit, (lo - rayO[a]) * invD[a]);
            }
            const float steps = (tExit - tMin) * invDt;     // last sample index at or before tExit
            if (steps > (float)i && steps < 2.0e9f) i = min((int)steps, nSteps - 1);
            continue;
        }
        // Present block: the voxel of sample i and of the next kRayBatch-1 samples that still
        // fall into this block are fetched together (their addresses do not depend on each
        // other, only the hit test is sequential), so a ray pays one memory latency per batch
        // instead of one per sample.  Samples are then classified strictly in order.
        float bt[kRayBatch];
        Voxel bs[kRayBatch];
        bool inBlock[kRayBatch];
#pragma unroll
        for (int j = 0; j < kRayBatch; ++j) {
            bt[j] = tMin + (float)(i + j) * dt;
            const float4 pj = mat4_mul(fp.T, dx * bt[j], dy * bt[j], bt[j], 1.0f);
            const int jx = kFastDiv ? world2voxel1_fast(pj.x, fp.voxelSize, rcpVoxel) : world2voxel1(pj.x, fp.voxelSize);
            const int jy = kFastDiv ? world2voxel1_fast(pj.y, fp.voxelSize, rcpVoxel) : world2voxel1(pj.y, fp.voxelSize);
            const int jz = kFastDiv ? world2voxel1_fast(pj.z, fp.voxelSize, rcpVoxel) : world2voxel1(pj.z, fp.voxelSize);
            inBlock[j] = (i + j < nSteps) && voxel2block1(jx) == kx && voxel2block1(jy) == ky &&
                         voxel2block1(jz) == kz;
            const int lx = (int)((uint32_t)jx - (uint32_t)kx * 8u);
            const int ly = (int)((uint32_t)jy - (uint32_t)ky * 8u);
            const int lz = (int)((uint32_t)jz - (uint32_t)kz * 8u);
            bs[j] = inBlock[j] ? dp.blocks[(size_t)cptr + (size_t)(lz * 64 + ly * 8 + lx)] : Voxel{0.0f, 0.0f};
        }
        bool done = false;
        int used = 0;
#pragma unroll
        for (int j = 0; j < kRayBatch; ++j) {
            if (done || !inBlock[j]) { done = true; continue; }   // first sample outside: back to the general path
            used = j + 1;
            if (!(bs[j].weight > 0.0f)) { prevValid = false; continue; }
            if (prevValid && prevSdf > 0.0f && bs[j].sdf <= 0.0f) {
                hit = prevT + (dt * prevSdf) / (prevSdf - bs[j].sdf);
                found = true;
                done = true;
                continue;
            }
            prevValid = true; prevSdf = bs[j].sdf; prevT = bt[j];
        }
        if (found) break;
        i += used - 1;            // sample i itself is always in the block, so used >= 1
    }
    depthOut[(size_t)v * fp.width + u] = hit;
}

// ---------------------------------------------------------------------------
// depth pre-processing (preProcess, CameraTrackingUtils.cu:115-120)
// ---------------------------------------------------------------------------
// calculateVertexPositions (:50-73) and calculateNormals (:75-113) as ONE kernel: the
// reference writes the vertex map, synchronises, and reads it back five times per pixel for
// the normals; here the four neighbour vertices are recomputed from the 2-byte depth (same
// arithmetic, same bits), so the pass reads 2 B and writes 32 B per pixel.
struct Mat3 { float m[9]; };

__device__ __forceinline__ float3 vertex_from_depth(const uint16_t *__restrict__ depth, const Mat3 &kinv, int W,
                                                    int x, int y)
{
    const float d = (float)depth[(size_t)y * W + x] / 5000.0f;          // :64, 5000 units = 1 m
    const float fx = (float)x, fy = (float)y;
    const float px = kinv.m[0] * fx + kinv.m[1] * fy + kinv.m[2] * 1.0f; // K_inv * (x, y, 1)  :71
    const float py = kinv.m[3] * fx + kinv.m[4] * fy + kinv.m[5] * 1.0f;
    const float pz = kinv.m[6] * fx + kinv.m[7] * fy + kinv.m[8] * 1.0f;
    return make_float3(px * d, py * d, pz * d);
}

__global__ __launch_bounds__(256) void preprocess_kernel(const uint16_t *__restrict__ depth, const Mat3 kinv, int W,
                                                         int H, float4 *__restrict__ positions,
                                                         float4 *__restrict__ normals)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= W * H) return;
    const int y = idx / W, x = idx - y * W;
    const float3 cc = vertex_from_depth(depth, kinv, W, x, y);
    positions[idx] = make_float4(cc.x, cc.y, cc.z, 1.0f);                                    // :73
    float4 n = make_float4(0.0f, 0.0f, 0.0f, 0.0f);                                          // :90
    if (x > 0 && x < W - 1 && y > 0 && y < H - 1) {                                          // :92
        const float3 pc = vertex_from_depth(depth, kinv, W, x, y + 1);
        const float3 cp = vertex_from_depth(depth, kinv, W, x + 1, y);
        const float3 mc = vertex_from_depth(depth, kinv, W, x, y - 1);
        const float3 cm = vertex_from_depth(depth, kinv, W, x - 1, y);
        if (cc.x != 0.0f && pc.x != 0.0f && cp.x != 0.0f && mc.x != 0.0f && cm.x != 0.0f) { // :100
            const float ax = pc.x - mc.x, ay = pc.y - mc.y, az = pc.z - mc.z;
            const float bx = cp.x - cm.x, by = cp.y - cm.y, bz = cp.z - cm.z;
            const float nx = ay * bz - az * by, ny = az * bx - ax * bz, nz = ax * by - ay * bx;   // cross
            const float l = __builtin_sqrtf(nx * nx + ny * ny + nz * nz);                          // length
            if (l > 0.0f) n = make_float4(nx / l, ny / l, nz / l, 0.0f);                     // :105-109
        }
    }
    normals[idx] = n;
}

// ---------------------------------------------------------------------------
// set-up kernels (deviceAllocate, VoxelUtils.cu:151-166)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void reset_table_kernel(VoxelEntry *table, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        VoxelEntry e;
        e.pos[0] = e.pos[1] = e.pos[2] = VH_POS_SENTINEL;
        e.ptr = VH_FREE_BLOCK;
        e.offset = 0;
        table[i] = e;
    }
}

__global__ __launch_bounds__(256) void reset_heap_kernel(uint32_t *heap, uint32_t n)
{
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) heap[i] = i;
}

// debug / known-answer hook: runs the scalar helpers on n points
__global__ void debug_eval_kernel(const FrameParams fp, const float4 *__restrict__ pts, int n, int32_t *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = pts[i];
    const int3_ b = world2block(p.x, p.y, p.z, fp.voxelSize);
    int sx, sy;
    project(fp.proj, p.x, p.y, p.z, sx, sy);
    int32_t *o = out + (size_t)i * 8;
    o[0] = b.x; o[1] = b.y; o[2] = b.z;
    o[3] = (int32_t)hash_block(b.x, b.y, b.z, fp.numBuckets);
    o[4] = block_in_frustum(fp, b.x, b.y, b.z) ? 1 : 0;
    o[5] = sx; o[6] = sy;
    o[7] = f2i_rz(p.w);
}

}  // namespace vh
