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
// planeOut (nullable): the claim half also leaves the camera z of every pixel in a packed float plane
// for launch 2 to gather from -- 4 bytes per pixel written once here instead of a 16-byte-strided
// gather from the vertex map there (C3, launch 2: 72 MB of traffic for 40 MB of algorithmic bytes).
template <int kKind, class In>
__global__ __launch_bounds__(256) void frame_scan_claim_kernel(const FrameParams fp, const DevPtrs dp, const In in,
                                                               uint32_t numEntries, uint32_t claimBlocks,
                                                               int parity, float *__restrict__ planeOut, uint32_t claimSpan,
                                                               uint32_t claimRatio)
{
    // The two roles are interleaved over the grid in proportion (block b is a claim block when
    // floor((b+1)*claim/total) steps): workgroups are dispatched roughly in index order, and
    // with all claim blocks in front a large image would fill the chip with latency-bound
    // pixel work before the first byte of the table is streamed.
    // The claim tiles end before the grid does (claimSpan < total; vh_api_frame.hip: claim_span): a claim
    // workgroup is a chain of dependent reads of ~4 us, and one dispatched among the last workgroups of a
    // 20 us launch is its tail.
    // (claim_index: the claim tiles 0 .. claimBlocks-1 spread evenly over workgroups 0 .. claimSpan-1 by a
    // multiply-high with claimRatio = ceil(claimBlocks * 2^32 / claimSpan) -- the host's numbers; a 64-bit
    // division per workgroup here cost the launch 2 us)
    const uint32_t total = gridDim.x;
    const bool inSpan = blockIdx.x < claimSpan;
    const uint32_t claimBefore = inSpan ? __umulhi(blockIdx.x, claimRatio) : claimBlocks;
    const uint32_t claimAfter = inSpan ? __umulhi(blockIdx.x + 1u, claimRatio) : claimBlocks;
    if (claimAfter != claimBefore) {
        // the latency-bound pixel waves issue first when they are ready, so they are off the compute
        // unit sooner (17.9 -> 17.6 us; raising the streaming waves instead cost 0.25 us)
        __builtin_amdgcn_s_setprio(3);
        claim_tile(fp, dp, in, claimBefore, kFusedCand + parity, kNoPending, planeOut);
    } else {
        flatten_tile<kKind>(fp, dp, numEntries, blockIdx.x - claimBefore,
                            CompactOut{kScanCount + parity, kScanCountB + parity, numEntries}, total - claimBlocks);
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
__device__ __forceinline__ void frame_commit_integrate(const FrameParams &fp, const DevPtrs &dp, const Depth &verts,
                                                       uint32_t commitBlocks, int parity)
{
    const int scanCount = dp.counters[kScanCount + parity];          // end A of the list; end B:
    const int scanCountB = dp.counters[kScanCountB + parity];
    if (blockIdx.x >= commitBlocks) {
        integrate_list(fp, dp, dp.compact, scanCount, (int)(blockIdx.x - commitBlocks), (int)(gridDim.x - commitBlocks), verts,
                       scanCountB, owned_entries(fp));
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
            dp.counters[kCompactCount] = scanCount + scanCountB + atomicAdd(dp.counters + kNewCount + parity, 0);
            dp.counters[kLastCandidates] = demanded;
            dp.counters[kScanCount + (parity ^ 1)] = 0;
            dp.counters[kScanCountB + (parity ^ 1)] = 0;
            dp.counters[kNewCount + (parity ^ 1)] = 0;
            dp.counters[kFusedCand + (parity ^ 1)] = 0;
            dp.counters[kCommitTicket] = 0;
        }
    }
}

template <class Depth>
__global__ __launch_bounds__(256) void frame_commit_integrate_kernel(const FrameParams fp, const DevPtrs dp,
                                                                     const Depth verts, uint32_t commitBlocks,
                                                                     int parity)
{
    frame_commit_integrate(fp, dp, verts, commitBlocks, parity);
}

// ---------------------------------------------------------------------------
// the pipelined frame: ONE launch per frame (option "pipeline", vh_integrate_batch)
// ---------------------------------------------------------------------------
// The two launches of a frame are ordered by data -- commit(i) needs every claim of frame i -- but the
// second one is a few microseconds of latency (counters -> candidate -> bucket -> heap -> block) that a
// kernel boundary on each side turns into a quarter of the frame.  Here the launch of frame i+1 also
// carries the deferred half of frame i:
//     launch i+1 = { claim(i+1) || walk(i+1) || commit(i) + integrate(i) }
// so the only ordering left is one kernel boundary per frame.  What makes this exact:
//   * claim(i+1) must judge every bucket as it will be AFTER commit(i), which runs concurrently.  It
//     never reads a slot that is being written: frame i's claim words (final, written by the previous
//     launch) say for every bucket who won it and which slot the entry goes to (claim_word: slot -> key,
//     f), so the insertion in flight is substituted from there (probe_and_claim, Pending);
//   * walk(i+1) must list the entries allocated after commit(i) and visible from pose i+1.  An entry
//     that commit(i) is inserting is skipped by the walk whatever it sees of it (same claim words) and
//     appended to frame i+1's list by commit(i) itself after the frustum test with pose i+1;
//   * integrate(i) gathers depth from a private copy of frame i's camera-z plane (float) or sensor
//     image (uint16) that claim(i) wrote, so the caller's buffer is only read by the launch of its own
//     frame: no lifetime requirement beyond vh_integrate's;
//   * claim words, candidate lists and compact lists alternate between two buffers, the per-frame
//     counters between three sets (filled by frame i+1, consumed by frame i, cleared for frame i+2; the
//     rotation simply continues across flushes, a flush clearing the two sets it does not consume);
//   * a frame that would insert more entries than the heap has free blocks is refused as a whole (all its
//     winners count as heap_exhausted and retry next frame): then claim(i+1) knows -- from two numbers
//     that are stable while the launch runs -- that nothing is in flight and reads the table as it is.
//     (vh_integrate without the pipeline serves as many winners as there are blocks.)
// (Round 3 measured the slimmer argument block the round-2 review asked for -- one FrameParams, one DevPtrs, and of the
// pending frame only its inverse pose, lock epoch and three alternating pointers: 0.55 instead of 0.8 KB, scalar spills
// 173-208 -> 129-137 -- side by side with this form on one box (tools/ab_commits.sh): C2 18.6 vs 18.0 us, C3 70.6 vs
// 70.3, C2 with band allocation 24.1 vs 23.9.  Slower, so the two-struct form stays; the spills sit in the role
// prologues, outside every loop.)
struct PipeArgs {
    uint32_t claimBlocks, walkBlocks, commitBlocks, integrateBlocks;
    uint32_t numEntries;
    int32_t setNew, setOld, setClear;      // counter sets: filled, consumed, cleared by this launch
    uint32_t hasNew, hasOld;               // first launch of a run: no old frame; flush launch: no new frame
    uint32_t walkIndexed;                  // flatten_variant 4: the walk role runs over the bucket-occupancy bitmap
    uint32_t claimPerWave;                 // the claim role takes a launch tile per WAVE (claim_tile_wave; the walk-free builds only)
    uint32_t claimSpan, claimRatio;        // claim tiles are interleaved with the first claimSpan - claimBlocks walk tiles
    float *planeNew;                       // private depth copies written by claim(new) ...
    uint16_t *rawNew;
    uint32_t *filter;                      // claim filters (vh_alloc.hip: pend_maybe), three of kPendFilterWords words
    uint32_t filtNew, filtOld, filtClear;  // word offsets of the filter the new frame fills / the pending frame filled / this launch clears
    int32_t doneTag;                       // overflow list: the pending frame's tag (lock epochs since creation), published when its commit phase ends
    uint32_t spinLimit;                    // ... and how many polls a workgroup waits for it (wait_commit_done)
#ifdef VH_DEBUG_SKIP_ROLES
    uint32_t skipRoles;                    // diagnostics build (option "debug_skip_roles"): bit r set = workgroups of role r return at once
#endif
                                           // (0 commit, 1 integrate, 2 claim, 3 walk): what the launch costs without them
};

// Serialised pipelined launches (overflow list): the claim and walk workgroups of frame i+1 wait here until the commit phase
// of frame i -- workgroups of the SAME grid, with the lowest indices, hence dispatched first -- has published its tag.
// "Dispatched first" is how the hardware dispatches today, not a guarantee, so the wait is bounded: after `limit` polls
// (~1.5 us each) the workgroup counts itself in kSpinTimeouts and gives up; the host sees the counter in vh_get_counters and
// runs that context's overflow-list frames as two launches from then on (the frame that timed out has lost the work of the
// workgroups that gave up: vh_counters.spin_timeouts > 0 is an error report, not a mode).
constexpr uint32_t kSpinLimitDefault = 1u << 20;
__device__ __forceinline__ bool wait_commit_done(int32_t *counters, int32_t tag, uint32_t limit)
{
    __shared__ int reached;
    if (threadIdx.x == 0) {
        int ok = 0;
        for (uint32_t it = 0; it < limit; ++it) {
            if (__hip_atomic_load(counters + kPipeCommitDone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == tag) { ok = 1; break; }
            __builtin_amdgcn_s_sleep(8);
        }
        if (!ok) atomicAdd(counters + kSpinTimeouts, 1);
        reached = ok;
        // the acquire: ONE cache invalidate per workgroup (its CU's vector cache, the XCD's non-coherent L2 lines), by the
        // wave that saw the tag; the other waves' loads come behind the barrier
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return reached != 0;
}

template <class In, class Depth, bool kBand, bool kSerial>
__device__ __forceinline__ void frame_pipelined(const FrameParams &fpNew, const DevPtrs &dpNew, const In &inNew,
                                                const FrameParams &fpOld, const DevPtrs &dpOld, const Depth &depthOld,
                                                const PipeArgs &a)
{
    int32_t *counters = dpNew.counters;
    // is frame i's commit phase inserting (live) or refusing everything?  Both numbers are stable while
    // this launch runs: the count of claimed buckets was final when the previous launch ended, the
    // free-block count was stored by its last commit workgroup (or, in a run's first launch, its first workgroup) in a word this
    // launch does not write.
    const int demandedOld = a.hasOld ? counters[kPipeCand + a.setOld] : 0;
    const int candOld = min(demandedOld, (int)dpOld.candCapacity);
    // With the overflow list an insertion in flight can change a chain link and a slot behind another bucket, which the
    // claim phase cannot substitute from the claim words alone.  Those frames are serialised INSIDE the launch instead: the
    // claim and walk workgroups of frame i+1 start when commit(i) has published its tag, and then read the table as it
    // is -- one launch per frame still, TSDF update(i) still beside walk(i+1); commit(i) serves as many winners as the heap
    // has blocks, like the unpipelined frame (no whole-frame refusal), and leaves its new entries for walk(i+1) to find.
    // (kSerial = the context has the list on: a build of its own -- the code of this launch is weighed by the microsecond:
    // the run-time form of this switch cost the C2 launch 0.7 us)
    constexpr bool serial = kSerial;
    const bool live = serial ? a.hasOld != 0u : a.hasOld && counters[kPipeHeapFree + a.setOld] >= counters[kPipeWinners + a.setOld];
    // Roles by workgroup index: [commit][integrate][claim and walk interleaved as in
    // frame_scan_claim_kernel]: frame i's deferred half runs first, at full width, then the table streams.
    // In-process A/B on C2 / C3 (launch time, us): this order with 512 integrate workgroups 18.8 / 88.9; with
    // 2048 of them (mostly idle, but dispatched before the first walk tile) 19.7 / 91.9; integrate and claim
    // workgroups interleaved among the walk tiles 20.5 / 93.7 (the latency-bound block updates then hold the
    // slots the stream needs); all claim tiles before the walk 20.4 / 93.0; the deferred half at the END of
    // the grid, where the walk drains (or commit first, integrate last) 21.5 / 105.9 against 21.0 / 94.6 for
    // this order on that box.  (Those orders were selectable at run time for a while; the run-time mapping
    // with its 64-bit divisions per workgroup cost every launch 2 us and is gone.)
    const uint32_t b = blockIdx.x;
    uint32_t role, index;                          // 0 commit, 1 integrate, 2 claim, 3 walk
    if (b < a.commitBlocks) { role = 0; index = b; }
    else if (a.walkIndexed) {
        // The walk-free frame streams nothing: every role is a chain of dependent reads plus arithmetic, and a chain that starts
        // late is the launch's tail.  The index walk's workgroups are few and their chain is the longest (bitmap -> buckets ->
        // frustum test -> list), so they come first, then the TSDF update, then the claim tiles; spread among the claim tiles
        // like the streaming walk they cost C2 10.8-11.2 us against 9.0-9.2, C3 28.1 against 27.6 (same box, generic build,
        // profiles/r05_index_walk_shapes.txt); their issue priority makes no difference.
        const uint32_t r = b - a.commitBlocks;
        if (r < a.walkBlocks) { role = 3; index = r; }
        else if (r < a.walkBlocks + a.integrateBlocks) { role = 1; index = r - a.walkBlocks; }
        else { role = 2; index = r - a.walkBlocks - a.integrateBlocks; }
    }
    else if (b < a.commitBlocks + a.integrateBlocks) { role = 1; index = b - a.commitBlocks; }
    else {
        // the claim tiles are spread over the first a.claimSpan of the claim + walk workgroups (multiply-high
        // by a.claimRatio = ceil(claimBlocks * 2^32 / claimSpan): no division in the kernel)
        const uint32_t r = b - a.commitBlocks - a.integrateBlocks;
        if (r < a.claimSpan) {
            const uint32_t before = __umulhi(r, a.claimRatio), after = __umulhi(r + 1u, a.claimRatio);
            if (after != before) { role = 2; index = before; } else { role = 3; index = r - before; }
        } else {
            role = 3; index = r - a.claimBlocks;
        }
    }
#ifdef VH_DEBUG_SKIP_ROLES
    if (a.skipRoles & (1u << role)) return;
#endif
    // first launch of a run (no frame in flight): nobody pops the heap during it, so its first workgroup
    // leaves the free-block count the NEXT launch will test frame i+1's insertions against
    if (!a.hasOld && b == 0u && threadIdx.x == 0) counters[kPipeHeapFree + a.setNew] = counters[kHeapCounter] + 1;
    if (role >= 2u) {
        // ---- frame i+1: claim || walk ----
        if (!a.hasNew) return;
        if (serial && a.hasOld) {
            // (the commit and integrate workgroups have the lowest indices of the grid: they are running or done when this one starts)
            if (!wait_commit_done(counters, a.doneTag, a.spinLimit)) return;
        }
        const Pending pend{a.hasOld && !serial ? dpOld.claim : nullptr, dpOld.candidates, fpOld.epoch, live,
                           serial ? -1 : kPipeWinners + a.setNew,
                           a.filter && a.hasOld && !serial ? a.filter + a.filtOld : nullptr,
                           a.filter && !serial ? a.filter + a.filtNew : nullptr};
        if (role == 2u) {
            __builtin_amdgcn_s_setprio(3);
#ifdef VH_DEBUG_SKIP_ROLES
            if (a.skipRoles & 16u) {           // diagnostics: the claim tiles end before their probes (what the probes' wait costs the launch)
                FrameParams fpN = fpNew;
                fpN.flags |= kFlagDebugNoProbe;
                if (!kBand && a.claimPerWave) {
                    const uint32_t tile = index * (256u / kWave) + threadIdx.x / kWave;
                    if (tile < num_tiles(fpN)) claim_tile_wave<In>(fpN, dpNew, inNew, tile, kPipeCand + a.setNew, pend, a.planeNew, a.rawNew);
                } else {
                    claim_tile<In, kBand>(fpN, dpNew, inNew, index, kPipeCand + a.setNew, pend, a.planeNew, a.rawNew, a.walkIndexed != 0u);
                }
                return;
            }
#endif
            if (!kBand && a.claimPerWave) {
                // (a launch tile per wave: the host sized the role at a quarter of the tiles, vh_api_frame.hip)
                const uint32_t tile = index * (256u / kWave) + threadIdx.x / kWave;
                if (tile < num_tiles(fpNew)) claim_tile_wave<In>(fpNew, dpNew, inNew, tile, kPipeCand + a.setNew, pend, a.planeNew, a.rawNew);
            } else {
                claim_tile<In, kBand>(fpNew, dpNew, inNew, index, kPipeCand + a.setNew, pend, a.planeNew, a.rawNew, a.walkIndexed != 0u);
            }
        } else if (a.walkIndexed) {
            flatten_index_tile(fpNew, dpNew, index, CompactOut{kPipeScan + a.setNew, kPipeScanB + a.setNew, a.numEntries},
                               pend);                                              // (opt-in: not the reference's walk)
        } else {
            flatten_tile_ballot(fpNew, dpNew, a.numEntries, index,
                                CompactOut{kPipeScan + a.setNew, kPipeScanB + a.setNew, a.numEntries}, pend);
        }
        return;
    }
    if (!a.hasOld) return;
    // the filter frame i+2 will fill: cleared here (nobody reads or writes it during this launch); a flush launch clears
    // the one a new frame would have filled as well, so that a run always starts on an empty one
    if (role == 0u && a.filter) {
        for (uint32_t t = index * 256u + threadIdx.x; t < kPendFilterWords; t += a.commitBlocks * 256u) {
            a.filter[a.filtClear + t] = 0u;
            if (!a.hasNew) a.filter[a.filtNew + t] = 0u;
        }
    }
    const int scanOld = counters[kPipeScan + a.setOld];
    if (role == 1u) {
        // ---- frame i: TSDF update of the blocks its walk (and commit(i-1)) listed ----
        integrate_list(fpOld, dpOld, dpOld.compact, scanOld, (int)index, (int)a.integrateBlocks, depthOld,
                       counters[kPipeScanB + a.setOld], a.numEntries);
        return;
    }
    // ---- frame i: commit ----
    __shared__ VoxelEntry newEntry;
    __shared__ int inserted;
    const int workers = max(1, min(candOld, (int)a.commitBlocks));
    if ((int)index >= workers) return;
    for (int i = (int)index; i < candOld; i += (int)a.commitBlocks) {
        if (threadIdx.x == 0) {
            inserted = 0;
            const int4 k = dpOld.candidates[i];
            if (live) {
                VoxelEntry e;
                inserted = commit_candidate(fpOld, dpOld, k, e, (uint32_t)i, false) ? 1 : 0;
                if (inserted) {
                    newEntry = e;
                    dpOld.compact[scanOld + atomicAdd(counters + kPipeNew + a.setOld, 1)] = e;
                    // what walk(i+1) would have listed had it seen the entry
                    if (a.hasNew && !serial && block_in_frustum(fpNew, e.pos[0], e.pos[1], e.pos[2]))
                        dpNew.compact[atomicAdd(counters + kPipeScan + a.setNew, 1)] = e;
                }
            } else {
                const uint32_t local = hash_block(k.x, k.y, k.z, fpOld.numBuckets) - fpOld.bucketLo;
                const unsigned long long w = dpOld.claim[local];
                if (claim_epoch(w) == fpOld.epoch && claim_slot(w) == (uint32_t)i) atomicAdd(counters + kHeapExhausted, 1);
            }
        }
        __syncthreads();
        if (inserted) integrate_block(fpOld, dpOld, newEntry, depthOld);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        __threadfence();
        const int ticket = atomicAdd(counters + kCommitTicket, 1);
        if (ticket == workers - 1) {
            counters[kCompactCount] = scanOld + counters[kPipeScanB + a.setOld] + atomicAdd(counters + kPipeNew + a.setOld, 0);
            counters[kLastCandidates] = demandedOld;
            counters[kPipeScan + a.setClear] = 0;
            counters[kPipeScanB + a.setClear] = 0;
            counters[kPipeNew + a.setClear] = 0;
            counters[kPipeCand + a.setClear] = 0;
            counters[kPipeWinners + a.setClear] = 0;
            if (!a.hasNew) {                       // flush launch: the set a new frame would have filled is unused: the next
                counters[kPipeScan + a.setNew] = 0;    // run starts on it (the host keeps rotating), and finds it empty
                counters[kPipeScanB + a.setNew] = 0;
                counters[kPipeNew + a.setNew] = 0;
                counters[kPipeCand + a.setNew] = 0;
                counters[kPipeWinners + a.setNew] = 0;
            }
            counters[kPipeHeapFree + a.setNew] = atomicAdd(counters + kHeapCounter, 0) + 1;   // what the next launch starts with
            counters[kCommitTicket] = 0;
            if (serial) {                          // every committer fenced before its ticket: the table is settled
                __threadfence();
                __hip_atomic_store(counters + kPipeCommitDone, a.doneTag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// kBand: the new frame allocates a truncation band (vh_set_alloc_band > 0); false = the reference's frame, without the band code
// kLean != 0: the option flags of both frames are known when the kernel is BUILT -- 1: only the 4-entries-per-lane walk
// (the default of a table inside the Infinity Cache), 2: that plus non-temporal walk loads (a table beyond it) -- and the walk
// is the reference's: everything the flags guard (overflow list, DDA band, TSDF-update variants, the other walk forms) is
// folded away by the compiler: 3.3 k instead of 6.8 k instructions; same box, same process: C2 18.35 -> 17.57 us, C3 69.4 ->
// 68.4.  The host picks this build when the context's flags are exactly those.  (Folding the semantics and a bucket size of
// 5 in as well: C2 17.2 but C3 70.2; one at a time: the semantics C2 17.9 / C3 68.4, the bucket size 17.8 / 75.2, the shard's
// bucket range 17.45 / 69.0 against 17.6 / 68.5 -- not done: what the compiler makes of a smaller kernel is not monotone.
// The same builds of the two-launch kernels: no difference (C2 16.27 + 4.45 us either way, C3 63.0 + 11.7 / 63.4 + 11.4).)
// (Eight waves per SIMD instead of the seven the 106 SGPRs allow -- __launch_bounds__(256, 8): 78 SGPRs, 129 instead of 100 of
// them spilled to VGPR lanes -- measured in round 4, same box: C2 17.69 -> 18.37 us, C3 69.0 -> 69.4, C2 with the band 22.9 ->
// 25.4.  The launch does not lack wave slots; the spill traffic lengthens the claim tiles.  profiles/r04_ab_pipe_waves.txt)
template <class In, class Depth, bool kBand, bool kSerial, int kLean>
__global__ __launch_bounds__(256) void frame_pipelined_kernel(FrameParams fpNew, const DevPtrs dpNew, const In inNew,
                                                              FrameParams fpOld, const DevPtrs dpOld,
                                                              const Depth depthOld, PipeArgs a)
{
    if (kLean != 0) {
        // (3 / 4: the same two with the ray-DDA band, VH_BAND_RAY_DDA -- builds that exist with kBand only)
        // (5 / 6: the first two with the occupancy-index walk in place of the reference's -- the walk-free frame, no band)
        // (7 / 8: 5 / 6 with a launch tile per wave in the claim role, claim_tile_wave -- the walk-free frame of a large image)
        constexpr uint32_t flags = (kLean == 2 || kLean == 4 || kLean == 6 || kLean == 8 ? (kFlagWalkShort | kFlagWalkNt) : kFlagWalkShort) |
                                   (kLean == 3 || kLean == 4 ? kFlagBandRayDda : 0u);
        fpNew.flags = flags;
        fpOld.flags = flags;
        a.walkIndexed = kLean >= 5 ? 1u : 0u;
    }
    a.claimPerWave = (kLean == 7 || kLean == 8) ? 1u : 0u;      // (builds of their own: carried as a run-time switch by builds 5 / 6 it
                                                                //  cost the 640x480 walk-free frame 8.9 -> 9.45 us)
    frame_pipelined<In, Depth, kBand, kSerial>(fpNew, dpNew, inNew, fpOld, dpOld, depthOld, a);
}

}  // namespace vh
