// vh_walk.hip -- flattenIntoBuffer: the walk over the VoxelEntry array and its variants.
// Part of libvoxelhash_hip.so (gfx950); included by vh_kernels.hip after vh_device.h.
#pragma once

namespace vh {

// ---------------------------------------------------------------------------
// flattenIntoBuffer
// ---------------------------------------------------------------------------
// Only `ptr` decides whether an entry is live, and all but a few thousand of the millions of
// entries are free.  Every lane reads just the ptr dword of its entries (stride 20 B: a wave
// instruction covers 1280 contiguous bytes, every fetched line is consumed across the loads in
// flight).  The rare live entries are re-read in full and frustum-tested; slots in the compact
// list are taken with one atomic per wave instruction that has a hit (ballot + popcount).  The
// reference also clears the whole compact table first (VoxelUtils.cu:757-758, its own TODO calls
// it redundant); that pass is dropped.
// The ptr loads are non-temporal when the table is larger than the 256 MiB Infinity Cache (kFlagWalkNt,
// set at creation, option "walk_nt"): C3 launch 1 76.2 -> 70.2 us and launch 2 13.5 -> 11.7 us (the voxel
// blocks stay cached); on a resident table (C2) they cost 2 %.  tools/micro/membw.hip is the ceiling probe:
// 419 MB read once at 5.9 TB/s with dwordx4 loads, 6.0 with this access shape, 7.0 with it non-temporal.
// Measured and removed in round 2: 16-byte-chunk loads with the ptr picked out of the
// chunk (+10 %), a per-lane hit count with one wave scan (+3 %), a "mask" form that stored allocation
// ballots for a second launch to consume (launch 1 -0.4 us, launch 2 +8.5 us), and an LDS-DMA form
// (global_load_lds_dwordx4, 5 KiB chunks per wave, ptr and live entries read back from LDS: C3 75.9 vs
// 69.5 us with non-temporal loads in both, C2 17.7 vs 17.1 us; one chunk per wave and twice the workgroups:
// C2 18.3 vs 17.0 us fused, 16.1 vs 15.8 us as a launch of its own -- at 15.8 us = 6.6 TB/s the plain walk
// already reads the 105 MB table as fast as the best form of the ceiling probe).
constexpr int kFlattenThreads = 256;
#ifndef VH_ENTRIES_PER_LANE
#define VH_ENTRIES_PER_LANE 8
#endif
constexpr int kEntriesPerLane = VH_ENTRIES_PER_LANE;   // tuning knob (make EXTRA=-DVH_ENTRIES_PER_LANE=n)
// option "flatten_variant" (values kept from round 1)
enum WalkKind : int { kWalkStridedBallot = 3, kWalkIndexed = 4 };      // (5, a persistent prefetching walk, lost in round 1 and is gone)

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

// Where the fused launches append the entries they list.  One counter takes ~90 returning atomics per
// microsecond, and every wave that finds visible entries needs one: on C3 (5 400 visible entries in a 60 us
// walk) that rate IS the walk's speed -- switching the compaction off took the walk from 65.9 to 58.8 us, two
// counters instead of one from 65.9 to 60.5 (eight: 60.3).  So the list has two ends: workgroups with an even
// tile index append upwards from compact[0] through counter A, the odd ones downwards from the last entry of
// the buffer through counter B (on another cache line).  The consumers inside the frame (integrate_list) read
// both ends; for everyone else the host folds end B behind end A first (compact_fold_kernel, fold_compact), so
// the boundary still sees the reference's dense list [0, occupied).  counterB < 0: one dense list (step API).
struct CompactOut {
    int counterA, counterB;
    uint32_t numEntries;        // entries of the compact buffer
};
__device__ __forceinline__ void compact_append(const DevPtrs &dp, const CompactOut &out, uint32_t tileIndex,
                                               unsigned long long mask, bool hit, const VoxelEntry &ent)
{
    const int lane = threadIdx.x & (kWave - 1);
    const bool toB = out.counterB >= 0 && (tileIndex & 1u);
    int base = 0;
    const int leaderLane = __ffsll((long long)mask) - 1;
    if (lane == leaderLane) base = atomicAdd(dp.counters + (toB ? out.counterB : out.counterA), __popcll(mask));
    base = __shfl(base, leaderLane);
    if (hit) {
        const uint32_t pos = (uint32_t)base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        dp.compact[toB ? out.numEntries - 1u - pos : pos] = ent;
    }
}

// strided walk with one ballot + atomic per unrolled entry slot (hits are rare on small scenes)
template <int kN>
__device__ __forceinline__ void walk_load_tile(const FrameParams &fp, const DevPtrs &dp, uint32_t numEntries,
                                               uint32_t tileIndex, int32_t (&ptrs)[kN])
{
    const uint32_t tile = tileIndex * (kFlattenThreads * kN);
    const int32_t *words = reinterpret_cast<const int32_t *>(dp.table);
    if (fp.flags & kFlagWalkNt) {
#pragma unroll
        for (int j = 0; j < kN; ++j) {
            const uint32_t e = tile + j * kFlattenThreads + threadIdx.x;
            ptrs[j] = (e < numEntries) ? __builtin_nontemporal_load(words + (size_t)e * kEntryDwords + 3) : VH_FREE_BLOCK;
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < kN; ++j) {
        const uint32_t e = tile + j * kFlattenThreads + threadIdx.x;
        ptrs[j] = (e < numEntries) ? words[(size_t)e * kEntryDwords + 3] : VH_FREE_BLOCK;
    }
}

// pend (pipelined frames only): the frame whose commit phase runs concurrently.  The entry it is
// inserting into a bucket (slot f of the bucket's claim word of that epoch) may or may not be visible
// yet; it is skipped here whatever the walk sees, and appended by that commit phase itself.
template <int kN>
__device__ __forceinline__ void walk_process_tile(const FrameParams &fp, const DevPtrs &dp, uint32_t tileIndex,
                                                  const int32_t (&ptrs)[kN], const CompactOut &out,
                                                  const Pending &pend = kNoPending)
{
    const uint32_t tile = tileIndex * (kFlattenThreads * kN);
    bool any = false;
#pragma unroll
    for (int j = 0; j < kN; ++j) any |= (ptrs[j] != VH_FREE_BLOCK);
    if (__ballot(any) == 0ull) return;
#pragma unroll
    for (int j = 0; j < kN; ++j) {
        bool hit = false;
        VoxelEntry ent;
        if (ptrs[j] != VH_FREE_BLOCK) {
            const uint32_t e = tile + j * kFlattenThreads + threadIdx.x;
            // the entry and the claim word of its bucket are fetched together (one round trip); what the read
            // of a slot that is being written returns is not looked at
            ent = dp.table[e];
            bool inFlight = false;
            if (pend.claim && pend.live) {
                const uint32_t b = e / fp.bucketSize;
                if (pend_maybe(pend, b)) {
                    const unsigned long long w = pend.claim[b];
                    inFlight = claim_epoch(w) == pend.epoch && claim_f(w) == e - b * fp.bucketSize;
                }
            }
            if (!inFlight) hit = block_in_frustum(fp, ent.pos[0], ent.pos[1], ent.pos[2]);   // VoxelUtils.cu:732
        }
        const unsigned long long mask = __ballot(hit);
        if (mask == 0ull) continue;
        compact_append(dp, out, tileIndex, mask, hit, ent);
    }
}

// Entries per lane of the frame's walk: 4 (kFlagWalkShort, set at creation; option "walk_entries" 4 | 8).
// Twice the workgroups with half the loads each stream faster than 8 per lane once a workgroup costs
// nothing before its first load (in-process, pipelined launch: C2 18.8 vs 19.4 us, C3 72.3 vs 75.4;
// 3 per lane 18.8 / 75.1, 6 per lane 19.1 / 74.7, 2 per lane 19.5 / 79.6).  The multi-camera walk of the
// sharded path and the view selection keep 8 (16.4 vs 16.5-16.9 us with 4).
template <int kN>
__device__ __forceinline__ void flatten_tile_ballot_n(const FrameParams &fp, const DevPtrs &dp, uint32_t numEntries,
                                                      uint32_t tileIndex, const CompactOut &out, const Pending &pend)
{
    int32_t ptrs[kN];
    walk_load_tile(fp, dp, numEntries, tileIndex, ptrs);
    walk_process_tile(fp, dp, tileIndex, ptrs, out, pend);
}
constexpr int kEntriesPerLaneShort = 4;
__device__ __forceinline__ void flatten_tile_ballot(const FrameParams &fp, const DevPtrs &dp, uint32_t numEntries,
                                                    uint32_t tileIndex, const CompactOut &out,
                                                    const Pending &pend = kNoPending)
{
    if (fp.flags & kFlagWalkShort) flatten_tile_ballot_n<kEntriesPerLaneShort>(fp, dp, numEntries, tileIndex, out, pend);
    else flatten_tile_ballot_n<kEntriesPerLane>(fp, dp, numEntries, tileIndex, out, pend);
}

// NOT the reference algorithm (opt-in, "walk_index"): instead of visiting every VoxelEntry,
// walk the bucket-occupancy bitmap (1 bit per bucket, maintained by the commit phase) and read
// only the buckets that hold entries.  One lane per 32-bucket word; the compact SET is the
// same, the bytes moved are numBuckets/8 + 100 per non-empty bucket instead of 20*N.
// pend (pipelined frames): a bucket's slot that the concurrent commit phase is filling is not looked at
// (nothing is allocated behind it: it is the bucket's first free slot) -- that entry is appended by the
// commit phase itself; whether this walk already sees the bucket's occupancy bit or not makes no difference
// (a bucket whose bit is still clear held nothing before).
//
// This form (round 1) follows one bucket per lane and loop round, slot after slot: a chain of dependent reads per set bit
// and one returning atomic per slot that has a hit; a wave is as long as its fullest word.  Kept for tables with the
// overflow list (holes in the buckets, chains behind them).
#ifndef VH_INDEX_WORDS
#define VH_INDEX_WORDS 4       // (same box, walk-free launch of C2 / C3 / C5table, generic build, index tiles first: 1 word per lane with a
                               //  reservation per wave 10.2 / 35.7 / 43.0 us; 2 words 9.3 / 31.8 / 39.8; 2 words, a reservation per workgroup 9.2 / 28.3 /
                               //  32.7; 4 words 9.2 / 27.6 / 29.5; 8 words 10.2 / 29.0 / 29.7 -- profiles/r05_index_walk_shapes.txt)
#endif
constexpr int kIndexWords = VH_INDEX_WORDS;       // bitmap words per lane of an index tile: a workgroup covers 256 * 4 * 32 = 32 768 buckets
__device__ __forceinline__ void flatten_index_tile_chain(const FrameParams &fp, const DevPtrs &dp, uint32_t tileIndex,
                                                         const CompactOut &out, const Pending &pend = kNoPending)
{
    const uint32_t owned = fp.bucketHi - fp.bucketLo;
    const uint32_t numWords = (owned + 31u) / 32u;
    for (int k = 0; k < kIndexWords; ++k) {
    const uint32_t w = (tileIndex * kIndexWords + (uint32_t)k) * kFlattenThreads + threadIdx.x;
    uint32_t bits = (w < numWords) ? dp.bucketBits[w] : 0u;
    while (__ballot(bits != 0u) != 0ull) {
        const bool have = bits != 0u;
        const uint32_t bucket = w * 32u + (have ? (uint32_t)__ffs((int)bits) - 1u : 0u);
        if (have) bits &= bits - 1u;
        bool more = have;                   // entries form a prefix of the bucket (not with the overflow list: holes)
        const bool holes = (fp.flags & kFlagOverflow) != 0u;
        uint32_t inFlight = ~0u;            // slot of this bucket the concurrent commit phase writes
        if (have && pend.claim && pend.live && pend_maybe(pend, bucket)) {
            const unsigned long long cw = pend.claim[bucket];
            if (claim_epoch(cw) == pend.epoch) inFlight = claim_f(cw);
        }
        for (uint32_t s = 0; s < fp.bucketSize; ++s) {
            VoxelEntry ent;
            bool hit = false;
            if (s == inFlight) more = false;
            if (more) {
                ent = dp.table[(size_t)bucket * fp.bucketSize + s];
                const bool live = ent.ptr != VH_FREE_BLOCK;
                more = live || holes;
                hit = live && block_in_frustum(fp, ent.pos[0], ent.pos[1], ent.pos[2]);
            }
            const unsigned long long mask = __ballot(hit);
            if (__ballot(more) == 0ull && mask == 0ull) break;
            if (mask == 0ull) continue;
            compact_append(dp, out, tileIndex, mask, hit, ent);
        }
    }
    }
}

// The same walk with the work dealt out evenly, every read in flight at once, and ONE list reservation per workgroup and
// round (round 5).  Measured first: the round-1 form cost the C2 launch 3.6 of its 10.9 us (a wave is as long as the fullest
// of its 64 words, a dependent read per slot); with that fixed the C3 launch stood at 29 us against 15 without the walk --
// 2 048 waves x one returning atomic each on the list's two counters, which take ~90 per microsecond each
// (profiles/r05_index_roles.txt).  So: a lane reads kIndexWords bitmap words (32 contiguous bytes), the wave's non-empty
// buckets go to a list in LDS (a prefix sum of the popcounts says where), the list is dealt to the lanes -- two buckets per
// lane and round, their first two slots and (behind the claim filter) the pending frame's claim word requested together --
// the four waves' hit counts meet in LDS and one lane reserves the list slots of all of them: C5table 512 workgroups = 256
// atomics per counter and frame.  Slots 2 and beyond are read only behind a live slot 1 (a bucket holding three entries:
// rare at any load the reference's 5-slot buckets work at) and appended per wave.
constexpr int kIndexQueue = 512;
// Where the index walk's hits go.  The single-camera frame: an entry inside the frame's frustum, onto one of the two ends
// of the frame's list.  (vh_shard.hip has the multi-camera one: the mask of the cameras that see the block, one dense list.)
struct IndexSinkFrame {
    const FrameParams &fp;
    CompactOut out;
    bool toB;
    __device__ __forceinline__ uint32_t judge(const VoxelEntry &e) const { return block_in_frustum(fp, e.pos[0], e.pos[1], e.pos[2]) ? 1u : 0u; }
    __device__ __forceinline__ int counter() const { return toB ? out.counterB : out.counterA; }
    __device__ __forceinline__ void emit(const DevPtrs &dp, uint32_t pos, const VoxelEntry &e, uint32_t) const
    {
        dp.compact[toB ? out.numEntries - 1u - pos : pos] = e;
    }
};

template <class Sink>
__device__ __forceinline__ void flatten_index_tile_to(const FrameParams &fp, const DevPtrs &dp, uint32_t tileIndex,
                                                      const Sink &sink, const Pending &pend)
{
    constexpr int kWaves = kFlattenThreads / kWave;
    __shared__ uint32_t queue_[kWaves][kIndexQueue];
    __shared__ int rounds_[kWaves], hits_[kWaves], base_;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    uint32_t *q = queue_[wave];
    const uint32_t owned = fp.bucketHi - fp.bucketLo;
    const uint32_t numWords = (owned + 31u) / 32u;
    // words w0 .. w0 + kIndexWords - 1 of this lane: contiguous (two 16-byte loads; the bitmap has a word of padding, but a
    // whole tile beyond the end does not: tested word by word)
    const uint32_t w0 = (tileIndex * kFlattenThreads + threadIdx.x) * kIndexWords;
    uint32_t bits[kIndexWords];
#pragma unroll
    for (int k = 0; k < kIndexWords; ++k) bits[k] = (w0 + (uint32_t)k < numWords) ? dp.bucketBits[w0 + k] : 0u;
    const bool pending = pend.claim != nullptr && pend.live;
    for (;;) {
        // ---- the pass's buckets: a lane contributes its set bits while the wave's list has room ----
        int cnt = 0;
#pragma unroll
        for (int k = 0; k < kIndexWords; ++k) cnt += __popc(bits[k]);
        int incl = cnt;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const int n = __shfl_up(incl, d);
            if (lane >= d) incl += n;
        }
        const int excl = incl - cnt;
        const int m = min(__builtin_amdgcn_readlane(incl, kWave - 1), kIndexQueue);
        int room = max(0, min(cnt, kIndexQueue - excl)), at = excl;
#pragma unroll
        for (int k = 0; k < kIndexWords; ++k) {
            while (room > 0 && bits[k] != 0u) {
                q[at++] = (w0 + (uint32_t)k) * 32u + (uint32_t)__ffs((int)bits[k]) - 1u;
                bits[k] &= bits[k] - 1u;
                --room;
            }
        }
        if (lane == 0) rounds_[wave] = (m + 2 * kWave - 1) / (2 * kWave);
        __syncthreads();
        int rounds = 0;
#pragma unroll
        for (int i = 0; i < kWaves; ++i) rounds = max(rounds, rounds_[i]);
        // ---- two buckets per lane and round; the same number of rounds for the four waves (one, normally) ----
        for (int r = 0; r < rounds; ++r) {
            const int base = r * 2 * kWave;
            VoxelEntry e[2][2];
            unsigned long long cw[2] = {0ull, 0ull};
            uint32_t bk[2];
            bool hv[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int idx = base + kWave * j + lane;
                hv[j] = idx < m;
                bk[j] = hv[j] ? q[idx] : 0u;
                const VoxelEntry *slot = dp.table + (size_t)bk[j] * fp.bucketSize;
                e[j][0].ptr = VH_FREE_BLOCK; e[j][1].ptr = VH_FREE_BLOCK;
                if (hv[j]) {
                    e[j][0] = slot[0];
                    if (fp.bucketSize > 1u) e[j][1] = slot[1];
                    if (pending && pend_maybe(pend, bk[j])) cw[j] = pend.claim[bk[j]];
                }
            }
            uint32_t hit[2][2];                    // what the sink sees in the entry (0: nothing)
            bool deeper[2];
            uint32_t inFlight[2];
            int nh = 0;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                inFlight[j] = (pending && claim_epoch(cw[j]) == pend.epoch) ? claim_f(cw[j]) : ~0u;
                const bool live0 = hv[j] && inFlight[j] != 0u && e[j][0].ptr != VH_FREE_BLOCK;
                const bool live1 = live0 && inFlight[j] != 1u && e[j][1].ptr != VH_FREE_BLOCK;
                hit[j][0] = live0 ? sink.judge(e[j][0]) : 0u;
                hit[j][1] = live1 ? sink.judge(e[j][1]) : 0u;
                deeper[j] = live1 && fp.bucketSize > 2u && inFlight[j] != 2u;
                nh += (hit[j][0] ? 1 : 0) + (hit[j][1] ? 1 : 0);
            }
            // one reservation for the workgroup's hits of this round
            int hincl = nh;
#pragma unroll
            for (int d = 1; d < kWave; d <<= 1) {
                const int n = __shfl_up(hincl, d);
                if (lane >= d) hincl += n;
            }
            if (lane == kWave - 1) hits_[wave] = hincl;
            __syncthreads();
            int before = 0, total = 0;
#pragma unroll
            for (int i = 0; i < kWaves; ++i) {
                const int h = hits_[i];
                before += i < wave ? h : 0;
                total += h;
            }
            if (total != 0) {          // (the same for every wave)
                if (threadIdx.x == 0) base_ = atomicAdd(dp.counters + sink.counter(), total);
                __syncthreads();
                const int start = base_;
                uint32_t pos = (uint32_t)(start + before + hincl - nh);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int k = 0; k < 2; ++k)
                        if (hit[j][k]) { sink.emit(dp, pos, e[j][k], hit[j][k]); ++pos; }
            }
            // slots 2 ..: only behind a live slot 1 (per wave)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bool more = deeper[j];
                if (__ballot(more) == 0ull) continue;
                for (uint32_t s = 2; s < fp.bucketSize; ++s) {
                    VoxelEntry ent;
                    uint32_t h = 0u;
                    if (s == inFlight[j]) more = false;
                    if (more) {
                        ent = dp.table[(size_t)bk[j] * fp.bucketSize + s];
                        more = ent.ptr != VH_FREE_BLOCK;
                        h = more ? sink.judge(ent) : 0u;
                    }
                    const unsigned long long mask = __ballot(h != 0u);
                    if (__ballot(more) == 0ull && mask == 0ull) break;
                    if (mask == 0ull) continue;
                    int start = 0;
                    const int leader = __ffsll((long long)mask) - 1;
                    if (lane == leader) start = atomicAdd(dp.counters + sink.counter(), __popcll(mask));
                    start = __shfl(start, leader);
                    if (h != 0u) sink.emit(dp, (uint32_t)start + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull)), ent, h);
                }
            }
            __syncthreads();           // (hits_ and base_ are rewritten by the next round)
        }
        bool left = false;
#pragma unroll
        for (int k = 0; k < kIndexWords; ++k) left |= bits[k] != 0u;
        if (!__syncthreads_or(left ? 1 : 0)) break;       // (also: the lists are refilled by the next pass)
    }
}

__device__ __forceinline__ void flatten_index_tile(const FrameParams &fp, const DevPtrs &dp, uint32_t tileIndex,
                                                   const CompactOut &out, const Pending &pend = kNoPending)
{
    if (fp.flags & kFlagOverflow) { flatten_index_tile_chain(fp, dp, tileIndex, out, pend); return; }
    flatten_index_tile_to(fp, dp, tileIndex, IndexSinkFrame{fp, out, out.counterB >= 0 && (tileIndex & 1u)}, pend);
}

// tileIndex: index of this workgroup among the `walkBlocks` workgroups doing the walk
template <int kKind>
__device__ __forceinline__ void flatten_tile(const FrameParams &fp, const DevPtrs &dp, uint32_t numEntries,
                                             uint32_t tileIndex, const CompactOut &out, uint32_t walkBlocks)
{
    if constexpr (kKind == kWalkIndexed)
        flatten_index_tile(fp, dp, tileIndex, out);
    else
        flatten_tile_ballot(fp, dp, numEntries, tileIndex, out);
}

template <int kKind>
__global__ __launch_bounds__(kFlattenThreads) void flatten_kernel(const FrameParams fp, const DevPtrs dp,
                                                                  uint32_t numEntries)
{
    flatten_tile<kKind>(fp, dp, numEntries, blockIdx.x, CompactOut{kCompactCount, -1, numEntries}, gridDim.x);
}

}  // namespace vh
