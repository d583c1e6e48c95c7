// vh_shard.hip -- the multi-camera frame on a bucket-range shard (multi-GPU).
// Part of libvoxelhash_hip.so (gfx950); included by vh_kernels.hip after vh_device.h.
#pragma once

namespace vh {

// A fused-generation bin: the frames of a batch are generated one launch after the other, each counting its records in a counter
// of its own (eight ints in the bin's last two records; batches of up to 8 frames), so the records of frame b lie between two prefix
// sums of the counters; header .w = kBinFrameMarks | batch says so, .x is unused, and the records stay out of the last two (capacity - 2).
constexpr int kBinFrameMarks = 0x40000000;

// ---------------------------------------------------------------------------
// multi-camera frame on a bucket-range shard
// ---------------------------------------------------------------------------
// phase 1 for key bins that arrived from the other ranks: bin b = bins[b*binStride..],
// record 0 = {count,0,0,0}, records 1..count = {x,y,z,rank}.  One lock epoch for all
// bins; rank = camera<<24 | launch rank, so cameras are served in order.
// (binIndex, part, parts): this workgroup handles every parts-th 256-record slice of the bin.
__device__ __forceinline__ void claim_bin_slice(const FrameParams &fp, const DevPtrs &dp,
                                                const int4 *__restrict__ bins, int32_t capacity, int32_t binStride,
                                                uint32_t binIndex, uint32_t part, uint32_t parts, int candCounter,
                                                const Pending &pend = kNoPending, int frame = -1)
{
    // frame >= 0: ONE bin per (source, batch) -- the records of all frames of the batch, each with its frame index where a
    // per-frame bin's record has the camera id (the camera IS the source: bin index); this launch serves `frame` only
    const int4 *bin = bins + (size_t)binIndex * binStride;
    const int4 head = bin[0];
    int n = head.x;
    const bool frameMarks = frame >= 0 && (head.w & kBinFrameMarks) != 0;      // (fused generation: per-frame counters in the last two records)
    const int room = frameMarks ? capacity - 3 : capacity - 1;
    int lo = 0, hiMark = 0;
    if (frameMarks) {
        const int4 c0 = bin[capacity - 2], c1 = bin[capacity - 1];
        const int cnt[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        n = 0;
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            if (f == frame) lo = n;
            n += cnt[f];
            if (f == frame) hiMark = n;
        }
    }
    if (n > room) {
        if (part == 0 && threadIdx.x == 0) atomicAdd(dp.counters + kBinOverflow, 1);
        n = room;
    }
    // A per-batch bin is filled by one generation launch per group of frames, one after the other, and the generator leaves a
    // mark behind each of them (bin_mark_kernel: header .y, .z = the count after the first and the second launch, .w = frames per
    // launch | marks << 8): the records of `frame` lie between two marks, and its launch reads only those (round 5: every launch
    // of a batch scanned the whole bin for its own frame's records, 0.8 us per launch at C2 size -- profiles/r04_sharded_*.txt).
    // No marks (.w == 0: bins not made by this generator, or more than three launches per batch): the whole bin.
    if (frameMarks) {
        lo = min(lo, n);
        n = min(hiMark, n);
    } else if (frame >= 0 && head.w != 0) {
        const int per = head.w & 0xff, marks = (head.w >> 8) & 0xff, g = frame / per;
        const int m0 = min(head.y, n), m1 = min(head.z, n);
        lo = g == 0 ? 0 : g == 1 ? m0 : m1;
        n = g < marks ? (g == 0 ? m0 : m1) : n;
        if (g > 2) lo = n;                                   // (cannot be: the generator writes no marks then)
    }
#ifdef VH_DEBUG_SKIP_ROLES
    if (fp.flags & kFlagDebugNoProbe) n = min(n, lo + ((n - lo) >> 5));      // diagnostics: 1/32 of the frame's records (what an acknowledgement channel would leave)
#endif
    for (int i = lo + (int)part * 256 + (int)threadIdx.x; i < n; i += (int)parts * 256) {
        const int4 k = bin[1 + i];
        uint32_t rank = (uint32_t)k.w;
        if (frame >= 0) {
            if ((rank >> kRankCameraShift) != (uint32_t)frame) continue;
            rank = (binIndex << kRankCameraShift) | (rank & ((1u << kRankCameraShift) - 1u));
        }
        const uint32_t h = hash_block(k.x, k.y, k.z, fp.numBuckets);
        if (h < fp.bucketLo || h >= fp.bucketHi) continue;
        probe_and_claim(fp, dp, k.x, k.y, k.z, h, rank, candCounter, pend);
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
template <int kN = kEntriesPerLane>
__device__ __forceinline__ void flatten_multi_tile(const FrameParams &fp, const DevPtrs &dp, uint32_t numEntries,
                                                   uint32_t tileIndex, int32_t numCams,
                                                   const float *__restrict__ packets, size_t packetStride,
                                                   int counter, const Pending &pend = kNoPending)
{
    const uint32_t tile = tileIndex * (kFlattenThreads * kN);
    int32_t ptrs[kN];
    walk_load_tile(fp, dp, numEntries, tileIndex, ptrs);
    uint32_t seen[kN];                      // cameras whose frustum holds entry j
    int myCount = 0;
#pragma unroll
    for (int j = 0; j < kN; ++j) {
        seen[j] = 0;
        if (ptrs[j] == VH_FREE_BLOCK) continue;
        const uint32_t e = tile + j * kFlattenThreads + threadIdx.x;
        const VoxelEntry ent = dp.table[e];
        // (pipelined frames: the entry the concurrent commit phase is writing is skipped whatever is seen of it --
        // that commit phase appends it itself, vh_walk.hip: walk_process_tile)
        if (pend.claim && pend.live) {
            const uint32_t b = e / fp.bucketSize;
            const unsigned long long w = pend.claim[b];
            if (claim_epoch(w) == pend.epoch && claim_f(w) == e - b * fp.bucketSize) continue;
        }
        seen[j] = camera_mask(fp, ent.pos, numCams, packets, packetStride);
        myCount += seen[j] != 0u;
    }
    int slot = reserve_compact_slots(dp, counter, myCount);
    if (slot < 0) return;
#pragma unroll
    for (int j = 0; j < kN; ++j) {
        if (seen[j] == 0u) continue;
        dp.compact[slot] = dp.table[tile + j * kFlattenThreads + threadIdx.x];
        dp.compactMask[slot] = seen[j];
        ++slot;
    }
}

// The occupancy-index walk (vh_walk.hip: flatten_index_tile_to) for all cameras of the step: the walk-free multi-camera frame
// (flatten_variant 4; NOT the reference's walk).  An entry goes to the dense list once, with the mask of the cameras that see it.
struct IndexSinkMulti {
    const FrameParams &fp;
    int32_t numCams;
    const float *__restrict__ packets;
    size_t packetStride;
    int counterIndex;
    __device__ __forceinline__ uint32_t judge(const VoxelEntry &e) const { return camera_mask(fp, e.pos, numCams, packets, packetStride); }
    __device__ __forceinline__ int counter() const { return counterIndex; }
    __device__ __forceinline__ void emit(const DevPtrs &dp, uint32_t pos, const VoxelEntry &e, uint32_t seen) const
    {
        dp.compact[pos] = e;
        dp.compactMask[pos] = seen;
    }
};

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
template <bool kSensor>
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
        if constexpr (kSensor) {
            const DepthSensor src{reinterpret_cast<const uint16_t *>(pk + kPacketHeaderU16), pk[32], pk[33], pk[34], pk[35]};
            dirty |= tsdf_update(fp, pk + 16, src, bx, by, bz, v.x, v.y);
            dirty |= tsdf_update(fp, pk + 16, src, bx + 1, by, bz, v.z, v.w);
        } else {
            const DepthPlane src{pk + kPacketHeader, 1};
            dirty |= tsdf_update(fp, pk + 16, src, bx, by, bz, v.x, v.y);
            dirty |= tsdf_update(fp, pk + 16, src, bx + 1, by, bz, v.z, v.w);
        }
    }
    if (dirty) *cell = v;
}

template <bool kSensor>
__global__ __launch_bounds__(256) void integrate_multi_kernel(const FrameParams fp, const DevPtrs dp,
                                                              int32_t numCams, const float *__restrict__ packets,
                                                              size_t packetStride)
{
    const int count = dp.counters[kCompactCount];
    for (int b = blockIdx.x; b < count; b += gridDim.x)
        integrate_block_multi<kSensor>(fp, dp, dp.compact[b], dp.compactMask[b], numCams, packets, packetStride);
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
                                                                     size_t packetStride, int parity,
                                                                     uint32_t claimSpan, uint32_t claimRatio, int binFrame)
{
    // (claim workgroups spread over the first claimSpan workgroups of the grid, as in frame_scan_claim_kernel)
    const uint32_t claimBlocks = numBins * partsPerBin;
    const bool inSpan = blockIdx.x < claimSpan;
    const uint32_t claimBefore = inSpan ? __umulhi(blockIdx.x, claimRatio) : claimBlocks;
    const uint32_t claimAfter = inSpan ? __umulhi(blockIdx.x + 1u, claimRatio) : claimBlocks;
    if (claimAfter != claimBefore)
        claim_bin_slice(fp, dp, bins, capacity, binStride, claimBefore / partsPerBin, claimBefore % partsPerBin,
                        partsPerBin, kFusedCand + parity, kNoPending, binFrame);
    else
        flatten_multi_tile(fp, dp, numEntries, blockIdx.x - claimBefore, numCams, packets, packetStride,
                           kScanCount + parity);
}

template <bool kSensor>
__global__ __launch_bounds__(256) void frame_multi_commit_integrate_kernel(const FrameParams fp, const DevPtrs dp,
                                                                           int32_t numCams,
                                                                           const float *__restrict__ packets,
                                                                           size_t packetStride,
                                                                           uint32_t commitBlocks, int parity)
{
    const int scanCount = dp.counters[kScanCount + parity];
    if (blockIdx.x >= commitBlocks) {
        for (int b = blockIdx.x - commitBlocks; b < scanCount; b += gridDim.x - commitBlocks)
            integrate_block_multi<kSensor>(fp, dp, dp.compact[b], dp.compactMask[b], numCams, packets, packetStride);
        return;
    }
    __shared__ VoxelEntry newEntry;
    __shared__ uint32_t newMask;
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
        if (inserted && newMask != 0u)
            integrate_block_multi<kSensor>(fp, dp, newEntry, newMask, numCams, packets, packetStride);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        __threadfence();
        const int ticket = atomicAdd(dp.counters + kCommitTicket, 1);
        if (ticket == workers - 1) {
            dp.counters[kCompactCount] = scanCount + atomicAdd(dp.counters + kNewCount + parity, 0);
            dp.counters[kLastCandidates] = demanded;
            dp.counters[kScanCount + (parity ^ 1)] = 0;
            dp.counters[kScanCountB + (parity ^ 1)] = 0;      // (end B of the single-camera frame's list, vh_walk.hip)
            dp.counters[kNewCount + (parity ^ 1)] = 0;
            dp.counters[kFusedCand + (parity ^ 1)] = 0;
            dp.counters[kCommitTicket] = 0;
        }
    }
}

// The multi-camera frame in ONE launch (vh_apply_frames_batch with the option "pipeline_shards", on by default): the
// launch of frame i+1 carries the deferred half of frame i, exactly as frame_pipelined_kernel does for the single-camera
// frame (vh_frame.hip, where the argument for its exactness is written out):
//     launch i+1 = { commit(i) + TSDF update(i) || claim the bins of frame i+1 || walk the shard for frame i+1's cameras }
// The two frames differ in their lock epoch, their bins and packets (both lie in the batch's receive buffers, which
// stay valid until the batch has been applied) and in the three buffers that alternate (claim words, candidate list,
// compact list with its camera masks); the poses travel inside the packets.  A batch of B frames is B + 1 launches.
// Fused key generation (round 5; vh_dist option "fused_generation"): the launch that applies frame b of one exchange also turns
// frame b of THIS rank's camera for a LATER exchange into keys binned by owner + its packet -- the work of
// generate_keys_sensor_batch_kernel (vh_alloc.hip: generate_keys_groups), as a role of 256-lane workgroups of this grid instead of
// a kernel of its own on a second stream, which cost the frame launches ~6 % by running beside them and ~3 % in the gaps two
// queues leave (profiles/r05_timeline_C2sharded.txt).  No band, sensor frames (the host checks).
#ifndef VH_FUSED_GEN_GROUPS
#define VH_FUSED_GEN_GROUPS 6        // (same box, sharded leg with one rank, frames/s: first form 2 groups 40.4 k, 4: 47.1 k, 8: 47.9 k, 12: 37.6 k,
                                     //  16: 33.6 k; with the owner computed without a division and the role's priority raised 6: 49.2-49.5 k,
                                     //  8: 49.2-49.3 k, 10: 43.6 k -- six keeps its distance from the cliff; the separate generation 45.5-45.7 k;
                                     //  profiles/r05_fused_generation_ab.txt)
#endif
#ifndef VH_FUSED_GEN_PRIO
#define VH_FUSED_GEN_PRIO 3        // (s_setprio of the generating workgroups: 49.0 -> 49.3 k frames/s, three runs each)
#endif
constexpr int kFusedGenGroups = VH_FUSED_GEN_GROUPS;            // 16x16 pixel tiles per generation workgroup (one returning atomic per owner for all of them)
struct GenJob {
    uint32_t blocks;                           // workgroups of the role (0: none in this launch)
    int32_t numShards, capacity, binStride;
    int4 *bins;                                // this rank's send bins of the later exchange: [owner][capacity]
    float *packet;                             // ... and its packet of this frame
    const uint16_t *depth;
    uint32_t rankBase;                         // frame index << kRankCameraShift (per-batch bins)
    float T[16], Tinv[16], k[9], unit;
    int4 *clearBins;                           // the send bins whose headers the LAST frame's job zeroes for the exchange after (or null)
    int32_t clearStride;
    int32_t frame, batch;                      // this frame's index in its batch: the role's last workgroup marks the frame's end in the bins
};


struct MultiPipeArgs {
    uint32_t claimBlocks, walkBlocks, commitBlocks, integrateBlocks;     // roles by workgroup index, in this order: commit, integrate, [generation,] claim/walk interleaved
    uint32_t partsPerBin, numBins, numEntries;
    int32_t capacity, binStride, numCams;
    int32_t binFrame;            // >= 0: the bins hold the whole batch, this launch claims the records of that frame
    int32_t doneTag;             // overflow list: the pending frame's tag, published when its commit phase ends (vh_frame.hip)
    uint32_t spinLimit;          // ... and how many polls a workgroup waits for it
    int32_t setNew, setOld, setClear;
    uint32_t hasNew, hasOld;
    uint32_t walkShort;          // the walk takes 4 instead of 8 entries per lane
    uint32_t walkIndexed;        // flatten_variant 4: the walk runs over the bucket-occupancy bitmap, its tiles ahead of the claim slices
    uint32_t claimSpan, claimRatio;
    uint32_t epochOld;
    const int4 *binsNew;
    const float *packetsNew, *packetsOld;
    size_t packetStride;
    unsigned long long *claimOld;
    int4 *candOld;
    VoxelEntry *compactOld;
    uint32_t *maskOld;
    uint32_t candCapacityOld;
    GenJob gen;
};

// (builds with the option flags folded in, as frame_pipelined_kernel has them, were measured here too: 4.2 k instead of 5.3 k
// instructions, but the launch 19.2-19.4 us against 19.1 on the world-1 sharded leg: not kept)
// kIndexed: the walk-free multi-camera frame (a.walkIndexed), a build of its own -- with the index walk as a run-time branch of the
// one build the default launch went from 18.7 to 20.8 us on the world-1 sharded leg (the code of these launches is weighed by the
// microsecond, vh_frame.hip).
template <bool kSensor, bool kSerial, bool kIndexed = false, bool kGen = false>
__global__ __launch_bounds__(256) void frame_multi_pipelined_kernel(const FrameParams fp, const DevPtrs dp, const MultiPipeArgs a)
{
    // (first launch of a run: workgroup 0 -- whatever its role -- leaves the free-block count the next launch tests frame i+1's insertions against)
    if (!a.hasOld && blockIdx.x == 0u && threadIdx.x == 0) dp.counters[kPipeHeapFree + a.setNew] = dp.counters[kHeapCounter] + 1;
    if constexpr (kGen) {
        // ---- generation role: the first a.gen.blocks workgroups behind commit + TSDF update ----
        const uint32_t g0 = a.commitBlocks + a.integrateBlocks;
        if (blockIdx.x >= g0 && blockIdx.x < g0 + a.gen.blocks) {
            const uint32_t g = blockIdx.x - g0;
            if (VH_FUSED_GEN_PRIO) __builtin_amdgcn_s_setprio(VH_FUSED_GEN_PRIO);
            FrameParams fg = fp;
#pragma unroll
            for (int i = 0; i < 16; ++i) { fg.T[i] = a.gen.T[i]; fg.Tinv[i] = a.gen.Tinv[i]; }
            SensorImage in;
            in.depth = a.gen.depth;
#pragma unroll
            for (int i = 0; i < 9; ++i) in.k[i] = a.gen.k[i];
            in.unit = a.gen.unit;
            float *pk = a.gen.packet;
            if (g == 0 && threadIdx.x < kPacketHeaderU16) {
                const int t = threadIdx.x;
                pk[t] = t < 16 ? a.gen.T[t] : t < 32 ? a.gen.Tinv[t - 16] : t == 32 ? a.gen.k[6] : t == 33 ? a.gen.k[7] : t == 34 ? a.gen.k[8] : a.gen.unit;
            }
            if (g == 0 && a.gen.clearBins && (int)threadIdx.x < a.gen.numShards) {    // (the exchange after this one starts from empty bins)
                int4 *nb = a.gen.clearBins + (size_t)threadIdx.x * a.gen.clearStride;
                nb[0] = make_int4(0, 0, 0, kBinFrameMarks | a.gen.batch);
                nb[a.gen.capacity - 2] = make_int4(0, 0, 0, 0);
                nb[a.gen.capacity - 1] = make_int4(0, 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < kFusedGenGroups; ++j) {            // the image itself, one pixel per lane
                const int idx = ((int)g * kFusedGenGroups + j) * 256 + (int)threadIdx.x;
                if (idx < fp.width * fp.height) reinterpret_cast<uint16_t *>(pk + kPacketHeaderU16)[idx] = in.depth[idx];
            }
            // (batches of up to 8 frames: per-frame counters in the bins' last two records, generate_keys_groups; the host checks)
            generate_keys_groups<SensorImage, 256, kFusedGenGroups>(fg, in, a.gen.numShards, a.gen.bins, a.gen.capacity - 2, a.gen.binStride, nullptr,
                                                                    a.gen.rankBase, g * (uint32_t)kFusedGenGroups, a.gen.frame);
            return;
        }
    }
    const uint32_t genBlocks = kGen ? a.gen.blocks : 0u;
    int32_t *counters = dp.counters;
    const int demandedOld = a.hasOld ? counters[kPipeCand + a.setOld] : 0;
    const int candOld = min(demandedOld, (int)a.candCapacityOld);
    // (overflow list: frames serialised inside the launch, as in frame_pipelined: claim and walk wait for commit(i)'s tag)
    constexpr bool serial = kSerial;
    const bool live = serial ? a.hasOld != 0u : a.hasOld && counters[kPipeHeapFree + a.setOld] >= counters[kPipeWinners + a.setOld];
    const uint32_t b = blockIdx.x;
    if (b >= a.commitBlocks + a.integrateBlocks + genBlocks) {
        // ---- frame i+1: claim its bins || walk the shard for its cameras ----
        if (!a.hasNew) return;
        if (serial && a.hasOld) {
            if (!wait_commit_done(counters, a.doneTag, a.spinLimit)) return;
        }
        const Pending pend{a.hasOld && !serial ? a.claimOld : nullptr, a.candOld, a.epochOld, live, serial ? -1 : kPipeWinners + a.setNew};
        const uint32_t r = b - a.commitBlocks - a.integrateBlocks - genBlocks;
        uint32_t before = a.claimBlocks, after = a.claimBlocks;
        if constexpr (kIndexed) {
            // (the walk-free frame: the index tiles first, as in frame_pipelined)
            if (r < a.walkBlocks) {
                flatten_index_tile_to(fp, dp, r, IndexSinkMulti{fp, a.numCams, a.packetsNew, a.packetStride, kPipeScan + a.setNew}, pend);
                return;
            }
            before = r - a.walkBlocks; after = before + 1u;
        }
        else { if (r < a.claimSpan) { before = __umulhi(r, a.claimRatio); after = __umulhi(r + 1u, a.claimRatio); } }
        if (after != before) {
            __builtin_amdgcn_s_setprio(3);
            claim_bin_slice(fp, dp, a.binsNew, a.capacity, a.binStride, before / a.partsPerBin, before % a.partsPerBin, a.partsPerBin,
                            kPipeCand + a.setNew, pend, a.binFrame);
        } else {
            // (4 entries per lane for a large shard, as the single-camera frame walks; 8 for the shards of many ranks)
            if (a.walkShort) flatten_multi_tile<kEntriesPerLaneShort>(fp, dp, a.numEntries, r - before, a.numCams, a.packetsNew, a.packetStride, kPipeScan + a.setNew, pend);
            else flatten_multi_tile(fp, dp, a.numEntries, r - before, a.numCams, a.packetsNew, a.packetStride, kPipeScan + a.setNew, pend);
        }
        return;
    }
    if (!a.hasOld) return;
    const int scanOld = counters[kPipeScan + a.setOld];
    if (b >= a.commitBlocks) {
        // ---- frame i: TSDF update of the blocks its walk (and commit(i-1)) listed, cameras in order ----
        for (int k = (int)(b - a.commitBlocks); k < scanOld; k += (int)a.integrateBlocks)
            integrate_block_multi<kSensor>(fp, dp, a.compactOld[k], a.maskOld[k], a.numCams, a.packetsOld, a.packetStride);
        return;
    }
    // ---- frame i: commit ----
    __shared__ VoxelEntry newEntry;
    __shared__ uint32_t newMask;
    __shared__ int inserted;
    const int workers = max(1, min(candOld, (int)a.commitBlocks));
    if ((int)b >= workers) return;
    for (int i = (int)b; i < candOld; i += (int)a.commitBlocks) {
        if (threadIdx.x == 0) {
            inserted = 0;
            newMask = 0u;
            const int4 k = a.candOld[i];
            if (live) {
                VoxelEntry e;
                inserted = commit_candidate(fp, dp, k, e, (uint32_t)i, false, a.epochOld, a.claimOld) ? 1 : 0;
                if (inserted) {
                    const uint32_t seen = camera_mask(fp, e.pos, a.numCams, a.packetsOld, a.packetStride);
                    newEntry = e;
                    newMask = seen;
                    if (seen != 0u) {                   // what walk(i) would have listed had it seen the entry
                        const int slot = scanOld + atomicAdd(counters + kPipeNew + a.setOld, 1);
                        a.compactOld[slot] = e;
                        a.maskOld[slot] = seen;
                    }
                    if (a.hasNew && !serial) {          // ... and what walk(i+1) would have: that walk skipped it
                        const uint32_t next = camera_mask(fp, e.pos, a.numCams, a.packetsNew, a.packetStride);
                        if (next != 0u) {
                            const int slot = atomicAdd(counters + kPipeScan + a.setNew, 1);
                            dp.compact[slot] = e;
                            dp.compactMask[slot] = next;
                        }
                    }
                }
            } else {
                const uint32_t local = hash_block(k.x, k.y, k.z, fp.numBuckets) - fp.bucketLo;
                const unsigned long long w = a.claimOld[local];
                if (claim_epoch(w) == a.epochOld && claim_slot(w) == (uint32_t)i) atomicAdd(counters + kHeapExhausted, 1);
            }
        }
        __syncthreads();
        if (inserted && newMask != 0u)
            integrate_block_multi<kSensor>(fp, dp, newEntry, newMask, a.numCams, a.packetsOld, a.packetStride);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        __threadfence();
        const int ticket = atomicAdd(counters + kCommitTicket, 1);
        if (ticket == workers - 1) {
            counters[kCompactCount] = scanOld + atomicAdd(counters + kPipeNew + a.setOld, 0);
            counters[kLastCandidates] = demandedOld;
            counters[kPipeScan + a.setClear] = 0;
            counters[kPipeScanB + a.setClear] = 0;
            counters[kPipeNew + a.setClear] = 0;
            counters[kPipeCand + a.setClear] = 0;
            counters[kPipeWinners + a.setClear] = 0;
            if (!a.hasNew) {
                counters[kPipeScan + a.setNew] = 0;
                counters[kPipeScanB + a.setNew] = 0;
                counters[kPipeNew + a.setNew] = 0;
                counters[kPipeCand + a.setNew] = 0;
                counters[kPipeWinners + a.setNew] = 0;
            }
            counters[kPipeHeapFree + a.setNew] = atomicAdd(counters + kHeapCounter, 0) + 1;
            counters[kCommitTicket] = 0;
            if (serial) {
                __threadfence();
                __hip_atomic_store(counters + kPipeCommitDone, a.doneTag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// Sensor-depth packets of up to kGenBatch frames of one camera (blockIdx.y = frame): header
// {pose, inverse, K_inv row 2, depth unit} and a straight copy of the uint16 image, 4 bytes per lane.
struct SensorFrames {
    float T[kGenBatch][16];
    float Tinv[kGenBatch][16];
    const uint16_t *depth[kGenBatch];
    float k6, k7, k8, unit;
};

__global__ __launch_bounds__(256) void write_packets_u16_kernel(const SensorFrames fr, int32_t numPixels,
                                                                float *__restrict__ packets, size_t packetFrameStride)
{
    const int b = blockIdx.y;
    float *pk = packets + packetFrameStride * b;
    if (blockIdx.x == 0 && threadIdx.x < kPacketHeaderU16) {
        const int t = threadIdx.x;
        pk[t] = t < 16 ? fr.T[b][t] : t < 32 ? fr.Tinv[b][t - 16] : t == 32 ? fr.k6 : t == 33 ? fr.k7 : t == 34 ? fr.k8 : fr.unit;
    }
    const uint32_t *src = reinterpret_cast<const uint32_t *>(fr.depth[b]);
    uint32_t *dst = reinterpret_cast<uint32_t *>(pk + kPacketHeaderU16);
    const int words = numPixels / 2;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < words; i += gridDim.x * 256) dst[i] = src[i];
}

// Behind generation launch `g` (0, 1) of a per-batch bin: the count so far becomes mark g of the header (claim_bin_slice above).
__global__ void bin_mark_kernel(int4 *bins, int32_t numShards, int32_t binStride, int32_t g, int32_t framesPerLaunch, int32_t marks)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= numShards) return;
    int4 *head = bins + (size_t)i * binStride;
    const int count = head->x;
    if (g == 0) head->y = count; else head->z = count;
    head->w = framesPerLaunch | (marks << 8);
}

// ... of a fused-generation bin: header {0, 0, 0, kBinFrameMarks | batch}, the eight frame counters zero
__global__ void prepare_bins_fused_kernel(int4 *bins, int32_t numShards, int32_t binStride, int32_t capacity, int32_t batch)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= numShards) return;
    int4 *nb = bins + (size_t)i * binStride;
    nb[0] = make_int4(0, 0, 0, kBinFrameMarks | batch);
    nb[capacity - 2] = make_int4(0, 0, 0, 0);
    nb[capacity - 1] = make_int4(0, 0, 0, 0);
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

}  // namespace vh
