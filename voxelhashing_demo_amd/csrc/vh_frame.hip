// vh_frame.hip -- the fused frame: SDF_Hashtable::integrate in two launches.
// Part of libvoxelhash_hip.so (gfx950); included by vh_kernels.hip after vh_device.h.
#pragma once

namespace vh {

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
        // the latency-bound pixel waves issue first when they are ready, so they are off the compute
        // unit sooner (17.9 -> 17.6 us; raising the streaming waves instead cost 0.25 us)
        __builtin_amdgcn_s_setprio(3);
        claim_pixel(fp, dp, verts, claimBefore * 256 + threadIdx.x, kFusedCand + parity);
    } else {
        flatten_tile<kKind>(fp, dp, numEntries, blockIdx.x - claimBefore, kScanCount + parity, total - claimBlocks);
    }
}

// The same frame straight from the uint16 sensor image (vh_integrate_depth): the claim half
// computes each pixel's vertex in place, the TSDF update reads the image (DepthSensor).  Default
// walk only.
__global__ __launch_bounds__(256) void frame_scan_claim_sensor_kernel(const FrameParams fp, const DevPtrs dp,
                                                                      const SensorImage in, uint32_t numEntries,
                                                                      uint32_t claimBlocks, int parity)
{
    const uint32_t total = gridDim.x;
    const uint32_t claimBefore = (uint32_t)(((uint64_t)blockIdx.x * claimBlocks) / total);
    const uint32_t claimAfter = (uint32_t)(((uint64_t)(blockIdx.x + 1u) * claimBlocks) / total);
    if (claimAfter != claimBefore) {
        __builtin_amdgcn_s_setprio(3);
        claim_pixel(fp, dp, in, claimBefore * 256 + threadIdx.x, kFusedCand + parity);
    } else {
        flatten_tile<kWalkStridedBallot>(fp, dp, numEntries, blockIdx.x - claimBefore, kScanCount + parity,
                                         total - claimBlocks);
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
        if (lane == 0) dp.allocMask[(size_t)tileIndex * (kEntriesPerLane * 4) + j * 4 + wave] = m;
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
    // only the workgroups that have a candidate to serve take part in the ticket (a release fence and
    // a returning atomic on one word per workgroup: 128 of them cost 0.8 us of a steady-state frame
    // that has a few dozen candidates)
    const int workers = max(1, min(n, (int)commitBlocks));
    if ((int)blockIdx.x >= workers) return;
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
        if (ticket == workers - 1) {
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
template <class Depth>
__device__ __forceinline__ void frame_commit_integrate(const FrameParams &fp, const DevPtrs &dp, const Depth &verts,
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
    // only the workgroups that have a candidate to serve take part in the ticket (a release fence and
    // a returning atomic on one word per workgroup: 128 of them cost 0.8 us of a steady-state frame
    // that has a few dozen candidates)
    const int workers = max(1, min(n, (int)commitBlocks));
    if ((int)blockIdx.x >= workers) return;
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
        if (ticket == workers - 1) {
            dp.counters[kCompactCount] = scanCount + atomicAdd(dp.counters + kNewCount + parity, 0);
            dp.counters[kLastCandidates] = n;
            dp.counters[kScanCount + (parity ^ 1)] = 0;
            dp.counters[kNewCount + (parity ^ 1)] = 0;
            dp.counters[kFusedCand + (parity ^ 1)] = 0;
            dp.counters[kCommitTicket] = 0;
        }
    }
}

__global__ __launch_bounds__(256) void frame_commit_integrate_kernel(const FrameParams fp, const DevPtrs dp,
                                                                     const float4 *__restrict__ verts,
                                                                     uint32_t commitBlocks, int parity)
{
    frame_commit_integrate(fp, dp, DepthPlane{reinterpret_cast<const float *>(verts) + 2, 4}, commitBlocks, parity);
}

__global__ __launch_bounds__(256) void frame_commit_integrate_sensor_kernel(const FrameParams fp, const DevPtrs dp,
                                                                            const SensorImage in,
                                                                            uint32_t commitBlocks, int parity)
{
    frame_commit_integrate(fp, dp, DepthSensor{in.depth, in.k[6], in.k[7], in.k[8], in.unit}, commitBlocks, parity);
}

}  // namespace vh
