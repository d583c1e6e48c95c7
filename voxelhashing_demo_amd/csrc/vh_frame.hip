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
//
// In: where a pixel's vertex comes from -- VertexMap (the float4 map of the reference's
// interface) or SensorImage (vh_integrate_depth: the uint16 image, vertices computed in place).
template <int kKind, class In>
__global__ __launch_bounds__(256) void frame_scan_claim_kernel(const FrameParams fp, const DevPtrs dp, const In in,
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
        claim_tile(fp, dp, in, claimBefore, kFusedCand + parity);
    } else {
        flatten_tile<kKind>(fp, dp, numEntries, blockIdx.x - claimBefore, kScanCount + parity, total - claimBlocks);
    }
}

// Launch 2: the first commitBlocks workgroups serve the candidates (one candidate per
// workgroup pass: lane 0 inserts, then all 256 lanes integrate the new block and it is
// appended to the compact list); the others stride over the entries the walk found.
// Only the commit workgroups take a ticket (a word that every workgroup of a large grid
// increments costs tens of microseconds): the last of them publishes the occupied count
// and clears the counter set of the other parity for the next frame.
// Depth: where the TSDF update reads a pixel's camera z -- DepthPlane on &verts[0].z or DepthSensor.
template <class Depth>
__global__ __launch_bounds__(256) void frame_commit_integrate_kernel(const FrameParams fp, const DevPtrs dp,
                                                                     const Depth verts, uint32_t commitBlocks,
                                                                     int parity)
{
    const int scanCount = dp.counters[kScanCount + parity];
    if (blockIdx.x >= commitBlocks) {
        for (int b = blockIdx.x - commitBlocks; b < scanCount; b += gridDim.x - commitBlocks)
            integrate_block(fp, dp, dp.compact[b], verts);
        return;
    }
    __shared__ VoxelEntry newEntry;
    __shared__ int inserted;
    const int demanded = dp.counters[kFusedCand + parity];
    const int n = min(demanded, (int)dp.candCapacity);
    // only the workgroups that have a candidate to serve take part in the ticket (a release fence and
    // a returning atomic on one word per workgroup: 128 of them cost 0.8 us of a steady-state frame
    // that has a few dozen candidates)
    const int workers = max(1, min(n, (int)commitBlocks));
    if ((int)blockIdx.x >= workers) return;
    for (int i = blockIdx.x; i < n; i += commitBlocks) {
        if (threadIdx.x == 0) {
            VoxelEntry e;
            inserted = commit_candidate(fp, dp, dp.candidates[i], e, (uint32_t)i) ? 1 : 0;
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
            dp.counters[kLastCandidates] = demanded;
            dp.counters[kScanCount + (parity ^ 1)] = 0;
            dp.counters[kNewCount + (parity ^ 1)] = 0;
            dp.counters[kFusedCand + (parity ^ 1)] = 0;
            dp.counters[kCommitTicket] = 0;
        }
    }
}

}  // namespace vh
