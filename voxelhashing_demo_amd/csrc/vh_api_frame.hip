// vh_api_frame.hip -- C-ABI, the frame: step-level entry points, the fused frame, raycast, block silhouettes.
// Included by vh_api.hip (same translation unit: shares fail(), VH_HIP, DeviceGuard, launch()).

// ---------------------------------------------------------------------------
// per-frame steps
// ---------------------------------------------------------------------------
extern "C" int vh_reset_mutexes(vh_context *c)
{
    if (!c) return fail(VH_ERR_INVALID_ARGUMENT, "null context");
    // The reference memsets 4*numBuckets bytes every frame (VoxelUtils.cu:146-149).  Claim words carry
    // the epoch in their top 10 bits, so starting a new epoch invalidates every lock at once.  After
    // kMaxClaimEpoch epochs the words are cleared for real and the epoch restarts at 1.
    if (c->fp.epoch >= kMaxClaimEpoch) {
        DeviceGuard guard(c->device);
        const int rc = flush_pending(c);
        if (rc != VH_OK) return rc;
        const size_t bytes = sizeof(unsigned long long) * (size_t)c->ownedBuckets;
        if (c->claimBuf[0]) {
            VH_HIP(hipMemsetAsync(c->claimBuf[0], 0, bytes, c->stream));
            VH_HIP(hipMemsetAsync(c->claimBuf[1], 0, bytes, c->stream));
        } else {
            VH_HIP(hipMemsetAsync(c->dp.claim, 0, bytes, c->stream));
        }
        c->fp.epoch = 0;
    }
    c->fp.epoch += 1;
    c->epochTotal += 1;
    return VH_OK;
}

static inline int grid_for(size_t n, int per_block) { return (int)((n + per_block - 1) / per_block); }
// 16x16 launch tiles of the depth image (one claim workgroup each, vh_alloc.hip)
static inline uint32_t host_num_tiles(const vh_context *c)
{
    return (uint32_t)((c->fp.width + 15) / 16) * (uint32_t)((c->fp.height + 15) / 16);
}

// Profiling: a start/stop event pair per dispatch, taken from a pool that vh_get_kernel_times (and
// accumulate_times below) refill; a profiled run of any length keeps at most kMaxTimedLaunches
// pairs alive (at that point the stream is synchronised once and the pending pairs are folded
// into the totals).
constexpr size_t kMaxTimedLaunches = 8192;
static int accumulate_times(vh_context *c);

template <typename K, typename... Args>
static int launch(vh_context *c, int phase, K kernel, dim3 grid, dim3 block, Args... args)
{
    if (!c->profiling) {
        hipLaunchKernelGGL(kernel, grid, block, 0, c->stream, args...);
        return VH_OK;
    }
    if (c->timed.size() >= kMaxTimedLaunches) {
        const int rc = accumulate_times(c);
        if (rc != VH_OK) return rc;
    }
    TimedLaunch t{phase, nullptr, nullptr};
    if (!c->eventPool.empty()) {
        t.start = c->eventPool.back().first;
        t.stop = c->eventPool.back().second;
        c->eventPool.pop_back();
    } else {
        VH_HIP(hipEventCreate(&t.start));
        const hipError_t e = hipEventCreate(&t.stop);
        if (e != hipSuccess) {
            (void)hipEventDestroy(t.start);
            return fail(VH_ERR_HIP, "hipEventCreate", e);
        }
    }
    hipExtLaunchKernelGGL(kernel, grid, block, 0, c->stream, t.start, t.stop, 0, args...);
    c->timed.push_back(t);
    return VH_OK;
}

template <class In>
static int launch_alloc(vh_context *c, const In &in)
{
    int rc = launch(c, kPhaseClaim, alloc_claim_kernel<In>, dim3(host_num_tiles(c)), dim3(256), c->fp, c->dp, in);
    if (rc != VH_OK) return rc;
    rc = launch(c, kPhaseCommit, alloc_commit_kernel, dim3(32), dim3(256), c->fp, c->dp);
    c->compactArmed = (rc == VH_OK);
    return rc;
}

// workgroups of the table walk: 2048 entries each
static uint32_t walk_blocks(const vh_context *c)
{
    if (c->flattenVariant == kWalkIndexed)       // one lane per 32-bucket word of the occupancy bitmap
        return (uint32_t)grid_for(((size_t)c->ownedBuckets + 31) / 32, kFlattenThreads * kIndexWords);
    return (uint32_t)grid_for(c->numEntries, kFlattenThreads * ((c->fp.flags & kFlagWalkShort) ? kEntriesPerLaneShort : kEntriesPerLane));
}

static int launch_flatten(vh_context *c)
{
    const dim3 grid(walk_blocks(c));
    if (c->flattenVariant == kWalkIndexed)
        return launch(c, kPhaseFlatten, flatten_kernel<kWalkIndexed>, grid, dim3(kFlattenThreads), c->fp, c->dp,
                      (uint32_t)c->numEntries);
    return launch(c, kPhaseFlatten, flatten_kernel<kWalkStridedBallot>, grid, dim3(kFlattenThreads), c->fp, c->dp,
                  (uint32_t)c->numEntries);
}

template <class Depth>
static int launch_integrate(vh_context *c, const Depth &depth)
{
    return launch(c, kPhaseIntegrate, integrate_kernel<Depth>, dim3(c->integrateGrid), dim3(256), c->fp, c->dp, depth);
}

extern "C" int vh_alloc_blocks(vh_context *c, const vh_float4 *verts, const vh_float4 *normals)
{
    // normals: loaded into a dead variable by the reference (VoxelUtils.cu:631); read only by the DDA band
    if (!c || !verts) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    if (c->epochTotal == 0) return fail(VH_ERR_INVALID_ARGUMENT, "vh_reset_mutexes must start the frame");
    if ((c->fp.flags & kFlagOverflow) && c->allocEpoch == c->epochTotal)
        return fail(VH_ERR_INVALID_ARGUMENT, "with the overflow list on, allocBlocks runs once per lock epoch "
                                             "(several cameras: vh_insert_bins / vh_apply_frames_batch)");
    c->allocEpoch = c->epochTotal;
    DeviceGuard guard(c->device);
    { const int frc = flush_pending(c); if (frc != VH_OK) return frc; }
    int rc = launch_alloc(c, VertexMap{reinterpret_cast<const float4 *>(verts), reinterpret_cast<const float4 *>(normals)});
    if (rc != VH_OK) return rc;
    VH_HIP(hipGetLastError());
    return VH_OK;
}

extern "C" int vh_flatten(vh_context *c, int32_t *occupied_out)
{
    if (!c) return fail(VH_ERR_INVALID_ARGUMENT, "null context");
    DeviceGuard guard(c->device);
    { const int frc = flush_pending(c); if (frc != VH_OK) return frc; }
    if (!c->compactArmed)
        VH_HIP(hipMemsetAsync(c->dp.counters + kCompactCount, 0, sizeof(int32_t), c->stream));   // :760
    c->compactArmed = false;
    c->occupiedCounter = kCompactCount;
    c->foldA = -1;                                   // (the step kernel writes one dense list)
    int rc = launch_flatten(c);
    if (rc != VH_OK) return rc;
    VH_HIP(hipGetLastError());
    if (occupied_out) {
        int32_t n = 0;
        VH_HIP(hipMemcpyAsync(&n, c->dp.counters + kCompactCount, sizeof n, hipMemcpyDeviceToHost, c->stream));
        VH_HIP(hipStreamSynchronize(c->stream));                                               // :765
        *occupied_out = n;
        c->params.numOccupiedBlocks = (uint32_t)n;
    }
    return VH_OK;
}

extern "C" int vh_integrate_depth_map(vh_context *c, const vh_float4 *verts)
{
    if (!c || !verts) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    DeviceGuard guard(c->device);
    { const int frc = settle(c); if (frc != VH_OK) return frc; }
    int rc = launch_integrate(c, vertex_depth(reinterpret_cast<const float4 *>(verts)));
    if (rc != VH_OK) return rc;
    VH_HIP(hipGetLastError());
    return VH_OK;
}

// How many of the claim + walk workgroups of a launch the claim tiles are spread over.  A claim workgroup is
// a chain of dependent reads (vertex -> bucket -> claim word) of about 4 us; interleaved uniformly over the
// whole grid, the ones dispatched last are the tail of the launch.  Ending them early by the share that
// chain has of the launch (estimated from the table bytes the walk streams at ~6 TB/s) removes the tail;
// squeezing them further starves the stream.  In-process A/B of the pipelined launch (4 entries per lane in
// the walk): C2 19.5 us uniform, 18.6 by this rule (77 %), 18.9 at 65 % or 85 %; C2 with band allocation (a
// claim tile then runs ~12 us) 29.0 uniform, 26.2 at 70 %, 24.3 at 50 %, 24.6 at 40 %, 26.6 at 30 %; C3, a
// 72 us launch: 72.5 uniform, 72.2 by the rule (94 %).
static uint32_t claim_span(const vh_context *c, uint32_t claimBlocks, uint32_t walkBlocks)
{
    const uint32_t total = claimBlocks + walkBlocks;
    const double walk_us = (double)c->numEntries * sizeof(VoxelEntry) / 6.0e6;     // bytes / (6 TB/s) in us
    // with band allocation a pixel demands several keys: ~2 us more per sample
    double chain_us = 4.0;
    if (c->fp.allocBand > 0.0f)
        chain_us += 2.0 * std::min(8.0, 2.0 * std::ceil((double)c->fp.allocBand / (4.0 * c->fp.voxelSize)));
    const double share = std::min(1.0, std::max(0.5, 1.0 - chain_us / std::max(walk_us, 1.0)));
    // (strictly more workgroups than claim tiles whenever there is a walk: claim_ratio below must stay < 2^32)
    return std::min(total, std::max<uint32_t>(claimBlocks + (walkBlocks ? 1u : 0u), (uint32_t)(share * total)));
}

// ceil(claimBlocks * 2^32 / span): floor(r * ratio / 2^32) steps from 0 to claimBlocks in unit steps over r = 0 .. span
static uint32_t claim_ratio(uint32_t claimBlocks, uint32_t span)
{
    if (span == 0 || claimBlocks >= span) return 0xffffffffu;       // (no walk workgroups: every workgroup claims; see callers)
    return (uint32_t)((((uint64_t)claimBlocks << 32) + span - 1) / span);
}

// ---------------------------------------------------------------------------
// pipelined frames (vh_frame.hip: frame_pipelined_kernel)
// ---------------------------------------------------------------------------
// second claim array / candidate list / compact list and the private depth copies, on first use
static int ensure_pipeline_buffers(vh_context *c)
{
    if (c->claimBuf[1]) return VH_OK;
    const size_t npix = (size_t)c->fp.width * c->fp.height;
    const size_t claimBytes = sizeof(unsigned long long) * (size_t)c->ownedBuckets;
    unsigned long long *claim2 = nullptr;
    int4 *cand2 = nullptr;
    VoxelEntry *compact2 = nullptr;
    uint32_t *filter = nullptr;
    float *plane[2] = {nullptr, nullptr};
    uint16_t *raw[2] = {nullptr, nullptr};
    hipError_t e = hipMalloc((void **)&claim2, claimBytes);
    if (e == hipSuccess) e = hipMalloc((void **)&cand2, sizeof(int4) * (size_t)c->candAllocated);
    if (e == hipSuccess) e = hipMalloc((void **)&compact2, sizeof(VoxelEntry) * c->numEntries);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) {
        e = hipMalloc((void **)&plane[i], sizeof(float) * npix);
        if (e == hipSuccess) e = hipMalloc((void **)&raw[i], sizeof(uint16_t) * npix);
    }
    if (e == hipSuccess) e = hipMemsetAsync(claim2, 0, claimBytes, c->stream);
    if (e == hipSuccess) e = hipMalloc((void **)&filter, sizeof(uint32_t) * 3 * kPendFilterWords);
    if (e == hipSuccess) e = hipMemsetAsync(filter, 0, sizeof(uint32_t) * 3 * kPendFilterWords, c->stream);
    if (e != hipSuccess) {
        if (filter) (void)hipFree(filter);
        if (claim2) (void)hipFree(claim2);
        if (cand2) (void)hipFree(cand2);
        if (compact2) (void)hipFree(compact2);
        for (int i = 0; i < 2; ++i) { if (plane[i]) (void)hipFree(plane[i]); if (raw[i]) (void)hipFree(raw[i]); }
        return fail(e == hipErrorOutOfMemory ? VH_ERR_OUT_OF_MEMORY : VH_ERR_HIP, "pipeline buffers", e);
    }
    c->claimBuf[0] = c->dp.claim;          c->claimBuf[1] = claim2;
    c->candBuf[0] = c->dp.candidates;      c->candBuf[1] = cand2;
    c->compactBuf[0] = c->dp.compact;      c->compactBuf[1] = compact2;
    for (int i = 0; i < 2; ++i) { c->planeBuf[i] = plane[i]; c->rawBuf[i] = raw[i]; }
    c->claimFilter = filter;
    c->pipeParity = 0;                     // dp.* currently alias set 0
    return VH_OK;
}

static DevPtrs pipe_view(const vh_context *c, int parity)
{
    DevPtrs d = c->dp;
    d.claim = c->claimBuf[parity];
    d.candidates = c->candBuf[parity];
    d.compact = c->compactBuf[parity];
    return d;
}

// With the overflow list a one-launch frame is serialised inside the launch: every claim / walk workgroup of frame i+1 waits
// for commit(i) and then ACQUIRES -- a cache invalidate (buffer_inv sc1), which this chip serves one at a time.  Measured in
// round 4 on C2 (6 320 such workgroups per launch, profiles/r04_ab_overflow_serial.txt): 185 us per launch with the
// invalidate in every wave, 53 us with one per workgroup (wait_commit_done), against 16.4 + 4.6 us for the two-launch frame
// -- ~5.5 ns per waiting workgroup against ~3 us for a second launch.  So the serialised form is taken only where it is the
// cheaper one: up to kSerialMaxWaiters waiting workgroups (option "pipeline_overflow" 1, the default; 0 never, 2 always).
constexpr uint32_t kSerialMaxWaiters = 512;
static bool serial_launch_pays(const vh_context *c, uint32_t waiters)
{
    if (!(c->fp.flags & kFlagOverflow)) return true;
    if (c->serialFallback || c->pipelineOverflow == 0) return false;
    return c->pipelineOverflow == 2 || waiters <= kSerialMaxWaiters;
}

// can this context run its frames pipelined right now?
static bool pipeline_applies(const vh_context *c)
{
    return c->pipeline && c->fusedFrame && (c->flattenVariant == kWalkStridedBallot || c->flattenVariant == kWalkIndexed) &&
           c->fp.bucketSize <= kMaxPipelinedBucket && !c->viewBlocks && serial_launch_pays(c, host_num_tiles(c) + walk_blocks(c));
}

// One launch: {claim || walk} of the new frame (in != nullptr) and {commit + integrate} of the pending one.
template <class In>
static int launch_pipelined(vh_context *c, const In *in, int newSensor, const float newK[4])
{
    const bool hasNew = in != nullptr, hasOld = c->pipePending;
    if (!hasNew && !hasOld) return VH_OK;
    int rc;
    if (!hasOld && (rc = ensure_pipeline_buffers(c)) != VH_OK) return rc;
    // buffers alternate and counter sets rotate from frame to frame, across flushes too (a flush leaves
    // the two sets it did not consume empty; at creation all three are)
    const int oldParity = c->pipeParity, newParity = oldParity ^ 1;
    const int setOld = c->pipeSet, setNew = (setOld + 1) % 3;
    PipeArgs a;
    const bool band = c->fp.allocBand > 0.0f;       // (the new frame's; the pending frame's claims are done)
    // (without a band a claim workgroup takes four launch tiles, one per wave: claim_tile_wave, vh_alloc.hip)
    a.walkBlocks = hasNew ? walk_blocks(c) : 0u;
    a.walkIndexed = c->flattenVariant == kWalkIndexed ? 1u : 0u;
    const bool serial = (c->fp.flags & kFlagOverflow) != 0u;
    // the lean builds (vh_frame.hip): no band, no list, the reference's walk, and both frames' option flags exactly the walk's
    int lean = 0;
    if (!serial) {                        // (with a band: the ray band only -- kFlagBandDda is a flag like the others)
        const uint32_t fo = hasOld ? c->pipeFp.flags : c->fp.flags;
        if (a.walkIndexed) {                // the walk-free frame (flatten_variant 4): builds of its own, without a band
            if (!band && c->fp.flags == kFlagWalkShort && fo == kFlagWalkShort) lean = 5;
            else if (!band && c->fp.flags == (kFlagWalkShort | kFlagWalkNt) && fo == (kFlagWalkShort | kFlagWalkNt)) lean = 6;
        }
        else if (c->fp.flags == kFlagWalkShort && fo == kFlagWalkShort) lean = 1;
        else if (c->fp.flags == (kFlagWalkShort | kFlagWalkNt) && fo == (kFlagWalkShort | kFlagWalkNt)) lean = 2;
        else if (band && c->fp.flags == (kFlagWalkShort | kFlagBandRayDda) && fo == c->fp.flags) lean = 3;
        else if (band && c->fp.flags == (kFlagWalkShort | kFlagWalkNt | kFlagBandRayDda) && fo == c->fp.flags) lean = 4;
    }
    // The walk-free frame of a large image: a claim workgroup takes four launch tiles, one per wave (claim_tile_wave, vh_alloc.hip:
    // a quarter of the waves, each with four pixels per lane).  C3 28.4 -> 26.7 us same box; a 640x480 frame prefers the tile per
    // workgroup (8.9 against 11.1 us: the longer chain per wave is its tail), and so does every frame under the reference's walk
    // (C2 18.9 -> 19.9-22.7 us, C3 72.0 -> 71.5): profiles/r05_claim_wave_tile_ab.txt.  Hence the size rule.
    a.claimPerWave = (hasNew && (lean == 5 || lean == 6) && host_num_tiles(c) > 2400u) ? 1u : 0u;
    if (a.claimPerWave) lean += 2;                                // builds 7 / 8
    a.claimBlocks = !hasNew ? 0u : a.claimPerWave ? (host_num_tiles(c) + 3u) / 4u : host_num_tiles(c);
    a.commitBlocks = hasOld ? (uint32_t)c->commitBlocks : 0u;
    a.integrateBlocks = hasOld ? (uint32_t)c->pipeIntegrateGrid : 0u;
    // (the walk-free frame of a large image: its TSDF update is on the critical path, not under a walk -- twice the workgroups:
    // C3 26.1 -> 25.1 us, while C2 prefers the 512 it has, 8.7 against 9.0; option "pipe_integrate_grid" sets the base)
    if (hasOld && a.walkIndexed && host_num_tiles(c) > 2400u) a.integrateBlocks *= 2u;
    a.numEntries = (uint32_t)c->numEntries;
    a.setNew = kPipeSetStride * setNew; a.setOld = kPipeSetStride * setOld; a.setClear = kPipeSetStride * ((setNew + 1) % 3);
    a.hasNew = hasNew; a.hasOld = hasOld;
    a.claimSpan = claim_span(c, a.claimBlocks, a.walkBlocks);
    a.claimRatio = claim_ratio(a.claimBlocks, a.claimSpan);
    a.planeNew = (hasNew && !newSensor) ? c->planeBuf[newParity] : nullptr;
    a.rawNew = (hasNew && newSensor) ? c->rawBuf[newParity] : nullptr;
    a.filter = c->claimFilter;
    a.filtNew = kPendFilterWords * (uint32_t)setNew; a.filtOld = kPendFilterWords * (uint32_t)setOld;
    a.filtClear = kPendFilterWords * (uint32_t)((setNew + 1) % 3);
    a.doneTag = c->pipeDoneTag;
    a.spinLimit = c->spinLimit ? c->spinLimit : kSpinLimitDefault;
#ifdef VH_DEBUG_SKIP_ROLES
    a.skipRoles = (uint32_t)c->debugSkipRoles;
#endif
    const DevPtrs dpNew = pipe_view(c, newParity), dpOld = pipe_view(c, oldParity);
    const dim3 grid(a.commitBlocks + a.integrateBlocks + a.claimBlocks + a.walkBlocks);
    In inNew{};
    if (hasNew) inNew = *in;
#define VH_LAUNCH_PIPELINED(DEPTH, BAND, SERIAL, LEAN) \
    launch(c, kPhaseFramePipelined, frame_pipelined_kernel<In, DEPTH, BAND, SERIAL, LEAN>, grid, dim3(256), c->fp, dpNew, inNew, c->pipeFp, dpOld, d, a)
#define VH_LAUNCH_PIPELINED_ANY(DEPTH) \
    (lean == 1 ? (band ? VH_LAUNCH_PIPELINED(DEPTH, true, false, 1) : VH_LAUNCH_PIPELINED(DEPTH, false, false, 1)) \
     : lean == 2 ? (band ? VH_LAUNCH_PIPELINED(DEPTH, true, false, 2) : VH_LAUNCH_PIPELINED(DEPTH, false, false, 2)) \
     : lean == 3 ? VH_LAUNCH_PIPELINED(DEPTH, true, false, 3) \
     : lean == 4 ? VH_LAUNCH_PIPELINED(DEPTH, true, false, 4) \
     : lean == 5 ? VH_LAUNCH_PIPELINED(DEPTH, false, false, 5) \
     : lean == 6 ? VH_LAUNCH_PIPELINED(DEPTH, false, false, 6) \
     : lean == 7 ? VH_LAUNCH_PIPELINED(DEPTH, false, false, 7) \
     : lean == 8 ? VH_LAUNCH_PIPELINED(DEPTH, false, false, 8) \
     : serial ? (band ? VH_LAUNCH_PIPELINED(DEPTH, true, true, 0) : VH_LAUNCH_PIPELINED(DEPTH, false, true, 0)) \
              : (band ? VH_LAUNCH_PIPELINED(DEPTH, true, false, 0) : VH_LAUNCH_PIPELINED(DEPTH, false, false, 0)))
    if (hasOld && c->pipeSensor) {
        const DepthSensor d{c->rawBuf[oldParity], c->pipeK[0], c->pipeK[1], c->pipeK[2], c->pipeK[3]};
        rc = VH_LAUNCH_PIPELINED_ANY(DepthSensor);
    } else {
        const DepthPlane d{c->planeBuf[oldParity], 1};
        rc = VH_LAUNCH_PIPELINED_ANY(DepthPlane);
    }
#undef VH_LAUNCH_PIPELINED_ANY
#undef VH_LAUNCH_PIPELINED
    if (rc != VH_OK) return rc;
    if (serial && hasNew && hasOld) c->serialQueued = true;       // (its claim / walk workgroups wait: check_spin_timeouts)
    if (hasOld) { c->foldA = kPipeScan + a.setOld; c->foldB = kPipeScanB + a.setOld; c->foldNew = kPipeNew + a.setOld; }
    if (hasNew) {
        c->pipePending = true;
        c->pipeFp = c->fp;
        c->pipeDoneTag = (int32_t)(c->epochTotal & 0x7fffffffu) | 0x40000000;      // (never 0, the counter's initial value)
        c->pipeSet = setNew;
        c->pipeParity = newParity;
        c->pipeSensor = newSensor;
        if (newK) std::memcpy(c->pipeK, newK, sizeof c->pipeK);
        c->dp.claim = dpNew.claim;               // the "current" buffers of everything that is not pipelined
        c->dp.candidates = dpNew.candidates;
        c->dp.compact = dpNew.compact;
    } else {
        c->pipePending = false;          // (pipeSet / pipeParity stay: the next run starts on the following set)
    }
    c->occupiedCounter = kCompactCount;
    c->compactArmed = false;
    return VH_OK;
}

static int flush_single_pending(vh_context *c)
{
    if (!c->pipePending) return VH_OK;
    const int rc = launch_pipelined<VertexMap>(c, nullptr, 0, nullptr);
    if (rc != VH_OK) return rc;
    VH_HIP(hipGetLastError());
    return VH_OK;
}

// (at most one of the two is pending: every entry point that starts a frame of one kind flushes the other --
// vh_integrate / vh_integrate_depth the multi-camera half, vh_apply_frames_batch the single-camera one)
static int flush_pending(vh_context *c)
{
    const int rc = flush_single_pending(c);
    return rc != VH_OK ? rc : flush_multi_pending(c);
}

// For whoever looks at the compact list from outside a frame (download, device pointers, the step-level TSDF
// update, collection, an explicit flush or synchronisation): the pending half first, then end B of the
// two-ended list (vh_walk.hip: CompactOut) behind end A -- the reference's dense list [0, occupied) -- and
// always in the buffer the context was created with (c->compactHome): pipelined frames alternate between two
// compact buffers, but a PtrContainer fetched once (as the reference does, VoxelUtils.cu:141-148) must stay
// good for the dense list after every later vh_flush / vh_synchronize.  Once no frame is pending the two
// buffers are interchangeable, so the home buffer becomes the current one again.
static int settle(vh_context *c)
{
    int rc = flush_pending(c);
    if (rc != VH_OK) return rc;
    VoxelEntry *home = c->compactHome;
    const bool away = c->dp.compact != home;
    if (c->foldA < 0 && !away) return VH_OK;
    rc = c->foldA >= 0 ? launch(c, kPhaseFlatten, compact_fold_kernel, dim3(64), dim3(256), c->dp, home, (uint32_t)c->numEntries,
                                c->foldA, c->foldB, c->foldNew)
                       : launch(c, kPhaseFlatten, compact_fold_kernel, dim3(64), dim3(256), c->dp, home, (uint32_t)c->numEntries,
                                c->occupiedCounter, -1, -1);
    c->foldA = -1;
    if (rc != VH_OK) return rc;
    if (away) {
        c->compactBuf[c->pipeParity ^ 1] = c->dp.compact;       // (compactBuf[] exist: only pipelined frames move dp.compact)
        c->compactBuf[c->pipeParity] = home;
        c->dp.compact = home;
    }
    VH_HIP(hipGetLastError());
    return VH_OK;
}

extern "C" int vh_flush(vh_context *c)
{
    VH_TRACE("vh_flush");
    if (!c) return fail(VH_ERR_INVALID_ARGUMENT, "null context");
    DeviceGuard guard(c->device);
    return settle(c);
}

// One frame (pose and lock epoch already set): In = where the claim phase reads a pixel's vertex,
// Depth = where the TSDF update reads a pixel's camera z.  Honours "fused_frame" and "flatten_variant".
template <int kKind, class In>
static int launch_scan_claim(vh_context *c, const In &in, uint32_t claimBlocks, uint32_t scanBlocks, float *planeOut)
{
    return launch(c, kPhaseFrameScanClaim, frame_scan_claim_kernel<kKind, In>, dim3(claimBlocks + scanBlocks),
                  dim3(256), c->fp, c->dp, in, (uint32_t)c->numEntries, claimBlocks, c->fusedParity, planeOut,
                  claim_span(c, claimBlocks, scanBlocks), claim_ratio(claimBlocks, claim_span(c, claimBlocks, scanBlocks)));
}

// the packed camera-z plane launch 1 leaves for launch 2 (vertex-map input only)
static inline float *fused_plane(vh_context *c, const VertexMap &)
{
    if (!c->fusedPlane && hipMalloc((void **)&c->fusedPlane, sizeof(float) * (size_t)c->fp.width * c->fp.height) != hipSuccess)
        c->fusedPlane = nullptr;                  // (out of memory: launch 2 gathers from the vertex map as before)
    return c->fusedPlane;
}
static inline float *fused_plane(vh_context *, const SensorImage &) { return nullptr; }
static inline DepthPlane plane_depth(const DepthPlane &fromVerts, float *plane) { return plane ? DepthPlane{plane, 1} : fromVerts; }
static inline DepthSensor plane_depth(const DepthSensor &d, float *) { return d; }

static inline int pipe_is_sensor(const VertexMap &) { return 0; }
static inline int pipe_is_sensor(const SensorImage &) { return 1; }
struct PipeK { float v[4]; };
static inline PipeK pipe_k(const VertexMap &) { return PipeK{{0, 0, 0, 0}}; }
static inline PipeK pipe_k(const SensorImage &s) { return PipeK{{s.k[6], s.k[7], s.k[8], s.unit}}; }

template <class In, class Depth>
static int run_frame(vh_context *c, const In &in, const Depth &depth)
{
    int rc;
    c->occupiedCounter = kCompactCount;
    if (pipeline_applies(c)) {
        rc = launch_pipelined(c, &in, pipe_is_sensor(in), pipe_k(in).v);
        if (rc != VH_OK) return rc;
        if (c->profiling) c->profiledFrames += 1;
        VH_HIP(hipGetLastError());
        return VH_OK;
    }
    if ((rc = flush_pending(c)) != VH_OK) return rc;
    if (c->fusedFrame) {
        // two launches: {claim || table walk}, then {commit + integrate}; see vh_frame.hip
        const uint32_t claimBlocks = host_num_tiles(c);
        const uint32_t scanBlocks = walk_blocks(c);
        // Large images: launch 1 also leaves the camera z of every pixel in a packed float plane and launch 2
        // gathers from that instead of the 16-byte-strided .z of the vertex map, which drags every line of the
        // map through launch 2 (C3: 72 MB of counter traffic for 40 MB of algorithmic bytes).  In-process A/B:
        // C3 (1.2 M pixels) launch 1 +1.4 us, launch 2 -2.9 us; C2 (0.3 M) +0.7 / -0.1 us: hence the size rule.
        float *plane = (size_t)c->fp.width * c->fp.height >= ((size_t)1 << 20) ? fused_plane(c, in) : nullptr;
        if (c->flattenVariant == kWalkIndexed)
            rc = launch_scan_claim<kWalkIndexed>(c, in, claimBlocks, scanBlocks, plane);
        else
            rc = launch_scan_claim<kWalkStridedBallot>(c, in, claimBlocks, scanBlocks, plane);
        if (rc != VH_OK) return rc;
        const uint32_t commitBlocks = (uint32_t)c->commitBlocks;
        rc = launch(c, kPhaseFrameCommitIntegrate, frame_commit_integrate_kernel<Depth>,
                    dim3(commitBlocks + (uint32_t)c->integrateGrid), dim3(256), c->fp, c->dp, plane_depth(depth, plane),
                    commitBlocks, c->fusedParity);
        if (rc != VH_OK) return rc;
        c->foldA = kScanCount + c->fusedParity; c->foldB = kScanCountB + c->fusedParity; c->foldNew = kNewCount + c->fusedParity;
        c->fusedParity ^= 1;       // this frame cleared the other counter set for the next one
        c->compactArmed = false;
    } else {
        // alloc_commit re-arms the compact counter, so no memset node is needed here
        c->foldA = -1;
        if ((rc = launch_alloc(c, in)) != VH_OK) return rc;
        c->compactArmed = false;
        if ((rc = launch_flatten(c)) != VH_OK) return rc;
        if ((rc = launch_integrate(c, depth)) != VH_OK) return rc;
    }
    if (c->profiling) c->profiledFrames += 1;
    VH_HIP(hipGetLastError());
    return VH_OK;
}

extern "C" int vh_integrate(vh_context *c, const float pose[16], const vh_float4 *verts, const vh_float4 *normals)
{
    VH_TRACE("vh_integrate");
    if (!c || !pose || !verts) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    DeviceGuard guard(c->device);
    // a single-camera frame on a context that still holds a multi-camera frame's deferred half (pipeline_shards 2, e.g. the
    // shard of a vh_dist): that half is served first -- the two pipelines share buffer parity and counter sets
    int rc = flush_multi_pending(c);
    if (rc == VH_OK) rc = vh_set_pose(c, pose);
    if (rc == VH_OK) rc = vh_reset_mutexes(c);
    if (rc != VH_OK) return rc;
    c->allocEpoch = c->epochTotal;
    const float4 *v = reinterpret_cast<const float4 *>(verts);
    return run_frame(c, VertexMap{v, reinterpret_cast<const float4 *>(normals)}, vertex_depth(v));
}

// The frame straight from the uint16 sensor image: preProcess's vertex computation happens inside
// the claim half, the TSDF update reads the image.  Equals vh_preprocess + vh_integrate.
extern "C" int vh_integrate_depth(vh_context *c, const float pose[16], const uint16_t *d_depth, const float k_inv[9])
{
    VH_TRACE("vh_integrate_depth");
    if (!c || !pose || !d_depth || !k_inv) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    DeviceGuard guard(c->device);
    int rc = flush_multi_pending(c);                      // (see vh_integrate)
    if (rc == VH_OK) rc = vh_set_pose(c, pose);
    if (rc == VH_OK) rc = vh_reset_mutexes(c);
    if (rc != VH_OK) return rc;
    SensorImage in;
    in.depth = d_depth;
    std::memcpy(in.k, k_inv, sizeof in.k);
    in.unit = 5000.0f;                                                   // CameraTrackingUtils.cu:64
    c->allocEpoch = c->epochTotal;
    return run_frame(c, in, DepthSensor{in.depth, in.k[6], in.k[7], in.k[8], in.unit});
}

// K frames in K + 1 launches: the pipeline switched on for the call, flushed at its end.
extern "C" int vh_integrate_batch(vh_context *c, int32_t count, const float *poses, const vh_float4 *const *d_verts,
                                  const vh_float4 *const *d_normals)
{
    if (!c || count < 0 || (count > 0 && (!poses || !d_verts))) return fail(VH_ERR_INVALID_ARGUMENT, "bad argument");
    const int saved = c->pipeline;
    c->pipeline = 1;
    int rc = VH_OK;
    for (int32_t i = 0; i < count && rc == VH_OK; ++i)
        rc = vh_integrate(c, poses + 16 * (size_t)i, d_verts[i], d_normals ? d_normals[i] : nullptr);
    c->pipeline = saved;
    if (!saved) {
        const int rc2 = vh_flush(c);
        if (rc == VH_OK) rc = rc2;
    }
    return rc;
}

extern "C" int vh_integrate_depth_batch(vh_context *c, int32_t count, const float *poses, const uint16_t *const *d_depth,
                                        const float k_inv[9])
{
    if (!c || count < 0 || (count > 0 && (!poses || !d_depth || !k_inv))) return fail(VH_ERR_INVALID_ARGUMENT, "bad argument");
    const int saved = c->pipeline;
    c->pipeline = 1;
    int rc = VH_OK;
    for (int32_t i = 0; i < count && rc == VH_OK; ++i) rc = vh_integrate_depth(c, poses + 16 * (size_t)i, d_depth[i], k_inv);
    c->pipeline = saved;
    if (!saved) {
        const int rc2 = vh_flush(c);
        if (rc == VH_OK) rc = rc2;
    }
    return rc;
}

// The raycast in the context's mode (option "raycast_mode"): the voxel DDA (default), optionally with the normal
// map of the hits written by the same pass, or the fixed-step march of rounds 1-2.
static int raycast_impl(vh_context *c, const float pose[16], float t_min, float t_max, float *d_depth_out,
                        vh_float4 *d_normals_out)
{
    VH_TRACE("vh_raycast");
    if (!c || !pose || !d_depth_out) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    if (!(t_max > t_min)) return fail(VH_ERR_INVALID_ARGUMENT, "t_max must exceed t_min");
    if (c->raycastMode == VH_RAYCAST_FIXED_STEP && d_normals_out)
        return fail(VH_ERR_INVALID_ARGUMENT, "the fixed-step march has no normal output (raycast_mode 0)");
    DeviceGuard guard(c->device);
    { const int frc = flush_pending(c); if (frc != VH_OK) return frc; }
    FrameParams fp = c->fp;
    std::memcpy(fp.T, pose, sizeof fp.T);
    dim3 grid((fp.width + 15) / 16, (fp.height + 15) / 16);
    DevPtrs dp = c->dp;
    if (c->viewBlocks) dp.blocks = const_cast<Voxel *>(c->viewBlocks);     // view table: voxels live in the records
    int rc;
    if (c->raycastMode == VH_RAYCAST_FIXED_STEP) {
        const float q = (t_max - t_min) / fp.voxelSize;
        int nsteps = (q >= 2147483648.0f) ? 0x7fffffff : (int)q;
        nsteps += 1;
        rc = launch(c, kPhaseRaycast, raycast_kernel, grid, dim3(256), fp, dp, c->rc_fx, c->rc_fy, c->rc_cx, c->rc_cy, t_min, nsteps,
                    d_depth_out);
    } else {
        RaycastArgs ra;
        ra.fx = c->rc_fx; ra.fy = c->rc_fy; ra.cx = c->rc_cx; ra.cy = c->rc_cy;
        ra.tMin = t_min; ra.tMax = t_max;
        float inv[16];
        invert4x4(pose, inv);                      // cofactor inverse (cuda_SimpleMatrixUtil.h:944-1069), as vh_set_pose
        ra.zrow[0] = inv[8] * fp.voxelSize; ra.zrow[1] = inv[9] * fp.voxelSize; ra.zrow[2] = inv[10] * fp.voxelSize;
        ra.zrow[3] = inv[11];
        // Hang guard, never reached by a ray: a ray changes voxel coordinate a at most (t_max - t_min) * |E_a| + 1
        // times, |E_a| <= (|T_a0| max|dx| + |T_a1| max|dy| + |T_a2|) / voxelSize.  A view whose bound is not finite
        // or beyond 2^22 steps is refused (the oracle walks at most that many voxels per ray).
        const double mdx = std::max(std::fabs((0.0 - ra.cx) / ra.fx), std::fabs(((double)fp.width - 1.0 - ra.cx) / ra.fx));
        const double mdy = std::max(std::fabs((0.0 - ra.cy) / ra.fy), std::fabs(((double)fp.height - 1.0 - ra.cy) / ra.fy));
        double steps = 16.0;
        for (int a = 0; a < 3; ++a)
            steps += 1.01 * ((double)t_max - (double)t_min) *
                     (std::fabs((double)pose[4 * a]) * mdx + std::fabs((double)pose[4 * a + 1]) * mdy + std::fabs((double)pose[4 * a + 2])) /
                     (double)fp.voxelSize + 2.0;
        if (!(steps < 4194304.0)) return fail(VH_ERR_INVALID_ARGUMENT, "view too deep for the voxel size (more than 2^22 voxel steps per ray) or not finite");
        ra.budget = (int)steps;
        // The camera centre in voxel-grid units, and the domain: inside an allocated block the kernel steps the voxel
        // coordinate as a float (exact below 2^24), so a view that can reach |coordinate| >= 2^23 is refused.
        double reach = 0.0;
        for (int a = 0; a < 3; ++a) {
            ra.G[a] = pose[4 * a + 3] / fp.voxelSize + 0.5f;
            reach = std::max(reach, std::fabs((double)ra.G[a]) + std::fabs((double)t_max) *
                                        (std::fabs((double)pose[4 * a]) * mdx + std::fabs((double)pose[4 * a + 1]) * mdy +
                                         std::fabs((double)pose[4 * a + 2])) / (double)fp.voxelSize + std::fabs((double)t_min) *
                                        (std::fabs((double)pose[4 * a]) * mdx + std::fabs((double)pose[4 * a + 1]) * mdy +
                                         std::fabs((double)pose[4 * a + 2])) / (double)fp.voxelSize);
        }
        if (!(reach < 8388608.0)) return fail(VH_ERR_INVALID_ARGUMENT, "view reaches beyond 2^23 voxels from the origin");
        ra.invVs = 1.0f / fp.voxelSize;
        ra.stamps = reinterpret_cast<unsigned long long *>(c->raycastStamps);
        // Which form: the cooperative one where ONE pass of its beam step covers the depth range -- 64 x kCoopSubs half-block slabs
        // (C2's 2 cm voxels: 5 m in one sub-pass; C3's 5 mm voxels: 4.9 m in four).  Round 6, with the lists shared inside the
        // workgroup (1280x960 views of the 5 mm model after 300 / 1 000 / 2 000 poses): 242 / 302 / 312 us against 238 / 319 / 335
        // for the per-lane walk behind the beam front end -- equal on a sparse model, ahead on a dense one (before the sharing the
        // per-lane walk was ahead at 5 mm: 151 against 175 us).  Longer ranges (several windows) keep the per-lane walk.
        int beam = c->raycastBeam;
        if (beam == 3) beam = (t_max - t_min) <= 64.0f * (float)kCoopSubs * 4.0f * fp.voxelSize ? 2 : 1;
        ra.beam = t_min > 0.0f ? beam : 0;
        float4 *nrm = reinterpret_cast<float4 *>(d_normals_out);
        const dim3 block(64 * kDdaBlockWaves);
        ra.patchesX = (fp.width + 7) / 8;
        ra.numPatches = ra.patchesX * ((fp.height + 7) / 8);
        ra.groups = (ra.numPatches + kDdaBlockWaves - 1) / kDdaBlockWaves;
        if (ra.beam == 2)       // the cooperative form: a workgroup per group of four far-apart patches (vh_raycast_coop.hip)
            rc = nrm ? launch(c, kPhaseRaycast, raycast_coop_kernel<true>, dim3((unsigned)ra.groups), block, fp, dp, ra, d_depth_out, nrm)
                     : launch(c, kPhaseRaycast, raycast_coop_kernel<false>, dim3((unsigned)ra.groups), block, fp, dp, ra, d_depth_out, nrm);
        else
            rc = nrm ? launch(c, kPhaseRaycast, raycast_dda_kernel<true>, grid, block, fp, dp, ra, d_depth_out, nrm)
                     : launch(c, kPhaseRaycast, raycast_dda_kernel<false>, grid, block, fp, dp, ra, d_depth_out, nrm);
    }
    if (rc != VH_OK) return rc;
    VH_HIP(hipGetLastError());
    return VH_OK;
}

extern "C" int vh_raycast(vh_context *c, const float pose[16], float t_min, float t_max, float *d_depth_out)
{
    return raycast_impl(c, pose, t_min, t_max, d_depth_out, nullptr);
}

extern "C" int vh_raycast_normals(vh_context *c, const float pose[16], float t_min, float t_max, float *d_depth_out,
                                  vh_float4 *d_normals_out)
{
    if (!d_normals_out) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    return raycast_impl(c, pose, t_min, t_max, d_depth_out, d_normals_out);
}

// Block silhouettes: SDFRenderer::drawToFrontAndBack, SDFRenderer.cpp:165-208.
extern "C" int vh_render_blocks(vh_context *c, const float pose[16], float t_min, float t_max, float *d_front,
                                float *d_back)
{
    VH_TRACE("vh_render_blocks");
    if (!c || !pose || !d_front || !d_back) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    if (!(t_max > t_min) || !(t_min >= 0.0f)) return fail(VH_ERR_INVALID_ARGUMENT, "need 0 <= t_min < t_max");
    DeviceGuard guard(c->device);
    { const int frc = flush_pending(c); if (frc != VH_OK) return frc; }
    BlockView bv;
    float inv[16];
    invert4x4(pose, inv);
    std::memcpy(bv.T, pose, sizeof bv.T);
    std::memcpy(bv.Tinv, inv, sizeof bv.Tinv);
    bv.fx = c->rc_fx; bv.fy = c->rc_fy; bv.cx = c->rc_cx; bv.cy = c->rc_cy;
    bv.tMin = t_min;
    bv.tMax = t_max;
    // records of the allocated blocks (32 bytes each + their 8-byte screen boxes on their own; a table holds at most
    // numVoxelBlocks of them, a view table one per entry) behind two counter words, allocated on first use (synchronises once)
    const size_t capacity = c->viewBlocks ? c->numEntries : std::min<size_t>(c->numEntries, c->params.numVoxelBlocks);
    if (!c->blockList || c->blockCapacity < capacity) {      // (a view context grows from 1 to numEntries at its first import)
        VH_HIP(hipStreamSynchronize(c->stream));
        if (c->blockList) (void)hipFree(c->blockList);
        c->blockList = nullptr;
        VH_HIP(hipMalloc((void **)&c->blockList, 16 + (sizeof(BlockRecord) + sizeof(uint2)) * capacity));
        VH_HIP(hipMemsetAsync(c->blockList, 0, 16, c->stream));
        c->blockCapacity = capacity;
        c->blockParity = 0;
    }
    int32_t *counts = c->blockList;
    BlockRecord *records = reinterpret_cast<BlockRecord *>(c->blockList + 4);
    uint2 *bounds = reinterpret_cast<uint2 *>(records + c->blockCapacity);
    const int parity = c->blockParity;
    const uint32_t words = (c->ownedBuckets + 31u) / 32u;
    int rc = launch(c, kPhaseRaycastBounds, blocks_list_kernel, dim3((unsigned)grid_for(words, 256)), dim3(256), c->fp, c->dp,
                    bv, records, bounds, (int32_t)capacity, counts, parity);
    if (rc == VH_OK)
        rc = launch(c, kPhaseRaycastBounds, blocks_tile_kernel, dim3((c->fp.width + 15) / 16, (c->fp.height + 15) / 16),
                    dim3(256), c->fp, bv, (const BlockRecord *)records, (const uint2 *)bounds, (int32_t)capacity, (const int32_t *)counts, parity,
                    d_front, d_back);
    c->blockParity ^= 1;
    if (rc != VH_OK) return rc;
    VH_HIP(hipGetLastError());
    return VH_OK;
}
