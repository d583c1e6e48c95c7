// vh_gc.hip -- block deletion and garbage collection.
// The reference's deleteVoxelEntry (VoxelUtils.cu:544-604) is unreachable and frees the block
// of the first FREE slot it meets; removeSingleBlockInHeap (:336-341) is its heap push.  Built
// as the paper the demo follows does it (Niessner et al. 2013, 4.4): identify the blocks of the
// compact list that hold nothing near a surface, remove their entries, zero their voxels and
// push them back on the heap.  Oracle: vho_delete_blocks / vho_garbage_collect.
// Part of libvoxelhash_hip.so (gfx950); included by vh_kernels.hip after vh_device.h.
#pragma once

namespace vh {

// Entries due for deletion are marked in a bitmap (one bit per entry, dp.gcMarks; all zero between
// calls).  (Round 1 kept the mark in VoxelEntry::offset, which now carries the overflow chain.)
__device__ __forceinline__ bool gc_marked(const DevPtrs &dp, uint32_t e)
{
    return (dp.gcMarks[e >> 5] >> (e & 31u)) & 1u;
}
__device__ __forceinline__ void gc_mark(const DevPtrs &dp, uint32_t e) { atomicOr(dp.gcMarks + (e >> 5), 1u << (e & 31u)); }
__device__ __forceinline__ void gc_unmark(const DevPtrs &dp, uint32_t e) { atomicAnd(dp.gcMarks + (e >> 5), ~(1u << (e & 31u))); }

// Marks the entry of `key` and puts its HOME bucket on the sweep list (once: the first marker of a
// bucket in this epoch swaps the consumed word into the bucket's claim word).
__device__ __forceinline__ void mark_for_deletion(const FrameParams &fp, const DevPtrs &dp, int kx, int ky, int kz)
{
    const uint32_t h = hash_block(kx, ky, kz, fp.numBuckets);
    if (h < fp.bucketLo || h >= fp.bucketHi) return;
    const uint32_t local = h - fp.bucketLo;
    uint32_t at = ~0u;
    if (fp.flags & kFlagOverflow) {
        uint32_t prev;
        at = find_entry_overflow(fp, dp.table, owned_entries(fp), local, kx, ky, kz, prev);
    } else {
        const VoxelEntry *bucket = dp.table + (size_t)local * fp.bucketSize;
        for (uint32_t s = 0; s < fp.bucketSize; ++s) {
            const VoxelEntry e = bucket[s];
            if (e.ptr == VH_FREE_BLOCK) break;                      // entries form a prefix
            if (e.pos[0] == kx && e.pos[1] == ky && e.pos[2] == kz) { at = local * fp.bucketSize + s; break; }
        }
    }
    if (at == ~0u) return;
    gc_mark(dp, at);                                                // idempotent: a key listed twice is freed once
    const unsigned long long tag = consumed_word(fp.epoch);
    if (atomicExch(dp.claim + local, tag) != tag)
        dp.compactMask[atomicAdd(dp.counters + kGcBuckets, 1)] = local;
}

// deleteVoxelEntry for a list of keys {x,y,z,_}
__global__ __launch_bounds__(256) void gc_mark_keys_kernel(const FrameParams fp, const DevPtrs dp,
                                                           const int4 *__restrict__ keys, int32_t n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int4 k = keys[i];
    mark_for_deletion(fp, dp, k.x, k.y, k.z);
}

// One compact entry per workgroup pass: max weight and min |sdf| over the observed voxels of
// its block (16 bytes per lane, wave reduction, then across the four waves through LDS);
// min/max are order independent, so the decision has the oracle's bits.
__global__ __launch_bounds__(256) void gc_identify_kernel(const FrameParams fp, const DevPtrs dp, int countIndex,
                                                          float threshold)
{
    __shared__ float sMin[4], sMax[4];
    const int count = dp.counters[countIndex];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    for (int b = blockIdx.x; b < count; b += gridDim.x) {
        const VoxelEntry e = dp.compact[b];
        const float4 v = *reinterpret_cast<const float4 *>(dp.blocks + (size_t)e.ptr + 2 * threadIdx.x);
        float mn = __builtin_inff(), mx = __builtin_fmaxf(v.y, v.w);
        if (v.y > 0.0f) mn = __builtin_fminf(mn, __builtin_fabsf(v.x));
        if (v.w > 0.0f) mn = __builtin_fminf(mn, __builtin_fabsf(v.z));
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            mn = __builtin_fminf(mn, __shfl_xor(mn, d));
            mx = __builtin_fmaxf(mx, __shfl_xor(mx, d));
        }
        if (lane == 0) { sMin[wave] = mn; sMax[wave] = mx; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const float m = __builtin_fminf(__builtin_fminf(sMin[0], sMin[1]), __builtin_fminf(sMin[2], sMin[3]));
            const float w = __builtin_fmaxf(__builtin_fmaxf(sMax[0], sMax[1]), __builtin_fmaxf(sMax[2], sMax[3]));
            if (w == 0.0f || m >= threshold) mark_for_deletion(fp, dp, e.pos[0], e.pos[1], e.pos[2]);
        }
        __syncthreads();
    }
}

// One lane per listed bucket: marked entries leave (their ptr goes to the freed list, which
// reuses the memory of the now stale compact list), the others move down in order so that the
// bucket's entries stay a prefix of its slots; the vacated tail becomes free slots.
__global__ __launch_bounds__(256) void gc_sweep_kernel(const FrameParams fp, const DevPtrs dp)
{
    const int n = dp.counters[kGcBuckets];
    int32_t *freed = reinterpret_cast<int32_t *>(dp.compact);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const uint32_t local = dp.compactMask[i];
        VoxelEntry *bucket = dp.table + (size_t)local * fp.bucketSize;
        uint32_t w = 0, s = 0;
        for (; s < fp.bucketSize; ++s) {
            const VoxelEntry e = bucket[s];
            if (e.ptr == VH_FREE_BLOCK) break;
            if (gc_marked(dp, local * fp.bucketSize + s)) {
                gc_unmark(dp, local * fp.bucketSize + s);
                freed[atomicAdd(dp.counters + kGcFreed, 1)] = e.ptr;
                continue;
            }
            if (w != s) bucket[w] = e;
            ++w;
        }
        VoxelEntry none;
        none.pos[0] = none.pos[1] = none.pos[2] = VH_POS_SENTINEL;
        none.ptr = VH_FREE_BLOCK;
        none.offset = 0;
        for (uint32_t k = w; k < s; ++k) bucket[k] = none;
        if (w == 0) atomicAnd(dp.bucketBits + (local >> 5), ~(1u << (local & 31u)));
        // (the macro-cell bitmap is hashed and shared: a stale bit only makes a ray skip less)
    }
}

// ---- the same with the overflow list on (oracle: delete_entry_overflow) --------------------------
// No compaction: a freed slot simply becomes free (Niessner et al. 2013, 4.2) -- except that a
// bucket's last slot, which heads its chain, is refilled with the first chained entry when it is
// deleted while the chain is not empty ("last slot free" always means "no chain"), and a chained
// entry is unlinked from its predecessor (prev.offset = curr.offset, VoxelUtils.cu:594).  The result
// does not depend on the order in which a set of keys is deleted, so one lane per listed HOME bucket
// can apply all of its deletions.  Two launches, because chained entries live in OTHER buckets' slots:
//   A  the marked entries in the home bucket's slots other than the last (told apart from foreign
//      chained entries by their hash; nobody writes those slots during A)
//   B  the last slot and the chain behind it (only this lane touches the chain's slots during B)
// the slot becomes free; the occupancy bit of its bucket is cleared when nothing lives there any more
// (a stale set bit is harmless: lookups and the index walk then read an empty bucket)
__device__ __forceinline__ void gc_reset_slot(const FrameParams &fp, const DevPtrs &dp, uint32_t e)
{
    VoxelEntry none;
    none.pos[0] = none.pos[1] = none.pos[2] = VH_POS_SENTINEL;
    none.ptr = VH_FREE_BLOCK;
    none.offset = 0;
    dp.table[e] = none;
    const uint32_t b = e / fp.bucketSize;
    bool any = false;
    for (uint32_t s = 0; s < fp.bucketSize; ++s) any |= dp.table[b * fp.bucketSize + s].ptr != VH_FREE_BLOCK;
    if (!any) atomicAnd(dp.bucketBits + (b >> 5), ~(1u << (b & 31u)));
}

__device__ __forceinline__ void gc_free_slot(const FrameParams &fp, const DevPtrs &dp, uint32_t e, int32_t *freed)
{
    freed[atomicAdd(dp.counters + kGcFreed, 1)] = dp.table[e].ptr;
    gc_reset_slot(fp, dp, e);
}

__global__ __launch_bounds__(256) void gc_sweep_overflow_a_kernel(const FrameParams fp, const DevPtrs dp)
{
    const int n = dp.counters[kGcBuckets];
    int32_t *freed = reinterpret_cast<int32_t *>(dp.compact);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const uint32_t local = dp.compactMask[i], start = local * fp.bucketSize;
        for (uint32_t s = 0; s + 1u < fp.bucketSize; ++s) {
            const uint32_t e = start + s;
            if (!gc_marked(dp, e)) continue;
            const VoxelEntry ent = dp.table[e];
            if (ent.ptr == VH_FREE_BLOCK || hash_block(ent.pos[0], ent.pos[1], ent.pos[2], fp.numBuckets) != local + fp.bucketLo)
                continue;                                           // a chained entry of another bucket: its home's lane frees it in B
            gc_unmark(dp, e);
            gc_free_slot(fp, dp, e, freed);
        }
    }
}

__global__ __launch_bounds__(256) void gc_sweep_overflow_b_kernel(const FrameParams fp, const DevPtrs dp)
{
    const int n = dp.counters[kGcBuckets];
    const uint32_t total = owned_entries(fp);
    int32_t *freed = reinterpret_cast<int32_t *>(dp.compact);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const uint32_t local = dp.compactMask[i], last = local * fp.bucketSize + fp.bucketSize - 1u;
        // the head: while it is marked, free its block and pull the next chained entry in
        while (dp.table[last].ptr != VH_FREE_BLOCK && gc_marked(dp, last)) {
            gc_unmark(dp, last);
            const VoxelEntry head = dp.table[last];
            if (head.offset == 0) {
                gc_free_slot(fp, dp, last, freed);
                break;
            }
            freed[atomicAdd(dp.counters + kGcFreed, 1)] = head.ptr;
            const uint32_t nx = chain_slot(last, head.offset, total);
            dp.table[last] = dp.table[nx];                          // pos, ptr and its link to the rest of the chain
            if (gc_marked(dp, nx)) { gc_unmark(dp, nx); gc_mark(dp, last); }
            gc_reset_slot(fp, dp, nx);
        }
        // the chain behind the (surviving) head: unlink the marked entries
        uint32_t prev = last;
        for (uint32_t iter = 0; iter < fp.listSize; ++iter) {
            const int32_t off = dp.table[prev].offset;
            if (off == 0 || dp.table[prev].ptr == VH_FREE_BLOCK) break;
            const uint32_t cur = chain_slot(last, off, total);
            if (gc_marked(dp, cur)) {
                gc_unmark(dp, cur);
                dp.table[prev].offset = dp.table[cur].offset;       // :594
                gc_free_slot(fp, dp, cur, freed);
            } else {
                prev = cur;
            }
        }
    }
}

// One freed block per workgroup pass: zero the 4 KiB (blocks are handed out zeroed) and push
// the block id back on the heap (removeSingleBlockInHeap, VoxelUtils.cu:336-341).
__global__ __launch_bounds__(256) void gc_release_kernel(const DevPtrs dp)
{
    const int n = dp.counters[kGcFreed];
    const int32_t *freed = reinterpret_cast<const int32_t *>(dp.compact);
    for (int b = blockIdx.x; b < n; b += gridDim.x) {
        const int32_t ptr = freed[b];
        reinterpret_cast<float4 *>(dp.blocks + (size_t)ptr)[threadIdx.x] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (threadIdx.x == 0) {
            const int addr = atomicAdd(dp.counters + kHeapCounter, 1);
            dp.heap[addr + 1] = (uint32_t)ptr / (uint32_t)kBlockVoxels;
        }
    }
}

// Closes a collection: totals, and the per-call counters and the (now stale) compact count go
// back to zero.
__global__ void gc_finish_kernel(const DevPtrs dp, int countIndex)
{
    const int n = dp.counters[kGcFreed];
    dp.counters[kFreedTotal] += n;
    dp.counters[kLastFreed] = n;
    dp.counters[kGcFreed] = 0;
    dp.counters[kGcBuckets] = 0;
    dp.counters[countIndex] = 0;
    dp.counters[kCompactCount] = 0;
}

}  // namespace vh
