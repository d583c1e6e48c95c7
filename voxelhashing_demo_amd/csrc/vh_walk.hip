// vh_walk.hip -- flattenIntoBuffer: the walk over the VoxelEntry array and its variants.
// Part of libvoxelhash_hip.so (gfx950); included by vh_kernels.hip after vh_device.h.
#pragma once

namespace vh {

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
#ifndef VH_ENTRIES_PER_LANE
#define VH_ENTRIES_PER_LANE 8
#endif
constexpr int kEntriesPerLane = VH_ENTRIES_PER_LANE;   // tuning knob (make EXTRA=-DVH_ENTRIES_PER_LANE=n)
constexpr int kChunksPerLane = kEntriesPerLane;
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

}  // namespace vh
