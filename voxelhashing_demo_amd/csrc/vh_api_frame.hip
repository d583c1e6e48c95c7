// vh_api_frame.hip -- C-ABI, the frame: step-level entry points, the fused frame, raycast, block silhouettes.
// Included by vh_api.hip (same translation unit: shares fail(), VH_HIP, DeviceGuard, launch()).

// ---------------------------------------------------------------------------
// per-frame steps
// ---------------------------------------------------------------------------
extern "C" int vh_reset_mutexes(vh_context *c)
{
    if (!c) return fail(VH_ERR_INVALID_ARGUMENT, "null context");
    // The reference memsets 4*numBuckets bytes every frame (VoxelUtils.cu:146-149).
    // Claim words carry the epoch in their upper half, so starting a new epoch
    // invalidates every lock at once.  After 2^32-1 frames the words are cleared
    // for real and the epoch restarts.
    if (c->fp.epoch == 0xffffffffu) {
        DeviceGuard guard(c->device);
        VH_HIP(hipMemsetAsync(c->dp.claim, 0, sizeof(unsigned long long) * (size_t)c->ownedBuckets, c->stream));
        c->fp.epoch = 0;
    }
    c->fp.epoch += 1;
    return VH_OK;
}

static inline int grid_for(size_t n, int per_block) { return (int)((n + per_block - 1) / per_block); }

template <typename K, typename... Args>
static int launch(vh_context *c, int phase, K kernel, dim3 grid, dim3 block, Args... args)
{
    if (!c->profiling) {
        hipLaunchKernelGGL(kernel, grid, block, 0, c->stream, args...);
        return VH_OK;
    }
    TimedLaunch t{phase, nullptr, nullptr};
    VH_HIP(hipEventCreate(&t.start));
    VH_HIP(hipEventCreate(&t.stop));
    hipExtLaunchKernelGGL(kernel, grid, block, 0, c->stream, t.start, t.stop, 0, args...);
    c->timed.push_back(t);
    return VH_OK;
}

static int launch_alloc(vh_context *c, const vh_float4 *verts)
{
    const int npix = c->fp.width * c->fp.height;
    int rc = launch(c, kPhaseClaim, alloc_claim_kernel, dim3(grid_for(npix, 256)), dim3(256), c->fp, c->dp,
                    reinterpret_cast<const float4 *>(verts));
    if (rc != VH_OK) return rc;
    rc = launch(c, kPhaseCommit, alloc_commit_kernel, dim3(32), dim3(256), c->fp, c->dp);
    c->compactArmed = (rc == VH_OK);
    return rc;
}

// workgroups of the table walk: 2048 entries each (strided) or 2048 16-byte chunks each (wide)
static uint32_t walk_blocks(const vh_context *c)
{
    if (c->flattenVariant == kWalkWide)
        return (uint32_t)grid_for(((size_t)c->numEntries * 20 + 15) / 16, kFlattenThreads * kChunksPerLane);
    if (c->flattenVariant == kWalkIndexed)       // one lane per 32-bucket word of the occupancy bitmap
        return (uint32_t)grid_for(((size_t)c->ownedBuckets + 31) / 32, kFlattenThreads);
    if (c->flattenVariant == kWalkPersistent)    // resident workgroups striding over the tiles
        return std::min<uint32_t>((uint32_t)grid_for(c->numEntries, kFlattenThreads * kEntriesPerLane),
                                  (uint32_t)c->persistentBlocks);
    return (uint32_t)grid_for(c->numEntries, kFlattenThreads * kEntriesPerLane);
}

static int launch_flatten(vh_context *c)
{
    const dim3 grid(walk_blocks(c));
    if (c->flattenVariant == kWalkIndexed)
        return launch(c, kPhaseFlatten, flatten_kernel<kWalkIndexed>, grid, dim3(kFlattenThreads), c->fp, c->dp,
                      (uint32_t)c->numEntries);
    if (c->flattenVariant == kWalkPersistent)
        return launch(c, kPhaseFlatten, flatten_kernel<kWalkPersistent>, grid, dim3(kFlattenThreads), c->fp, c->dp,
                      (uint32_t)c->numEntries);
    if (c->flattenVariant == kWalkWide)
        return launch(c, kPhaseFlatten, flatten_kernel<kWalkWide>, grid, dim3(kFlattenThreads), c->fp, c->dp,
                      (uint32_t)c->numEntries);
    if (c->flattenVariant == kWalkStrided)
        return launch(c, kPhaseFlatten, flatten_kernel<kWalkStrided>, grid, dim3(kFlattenThreads), c->fp, c->dp,
                      (uint32_t)c->numEntries);
    if (c->flattenVariant == kWalkStridedNT)
        return launch(c, kPhaseFlatten, flatten_kernel<kWalkStridedNT>, grid, dim3(kFlattenThreads), c->fp, c->dp,
                      (uint32_t)c->numEntries);
    return launch(c, kPhaseFlatten, flatten_kernel<kWalkStridedBallot>, grid, dim3(kFlattenThreads), c->fp, c->dp,
                  (uint32_t)c->numEntries);
}

static int launch_integrate(vh_context *c, const vh_float4 *verts)
{
    return launch(c, kPhaseIntegrate, integrate_kernel, dim3(c->integrateGrid), dim3(256), c->fp, c->dp,
                  reinterpret_cast<const float4 *>(verts));
}

extern "C" int vh_alloc_blocks(vh_context *c, const vh_float4 *verts, const vh_float4 *normals)
{
    (void)normals;   // loaded into a dead variable by the reference (VoxelUtils.cu:631)
    if (!c || !verts) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    if (c->fp.epoch == 0) return fail(VH_ERR_INVALID_ARGUMENT, "vh_reset_mutexes must start the frame");
    DeviceGuard guard(c->device);
    int rc = launch_alloc(c, verts);
    if (rc != VH_OK) return rc;
    VH_HIP(hipGetLastError());
    return VH_OK;
}

extern "C" int vh_flatten(vh_context *c, int32_t *occupied_out)
{
    if (!c) return fail(VH_ERR_INVALID_ARGUMENT, "null context");
    DeviceGuard guard(c->device);
    if (!c->compactArmed)
        VH_HIP(hipMemsetAsync(c->dp.counters + kCompactCount, 0, sizeof(int32_t), c->stream));   // :760
    c->compactArmed = false;
    c->occupiedCounter = kCompactCount;
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
    int rc = launch_integrate(c, verts);
    if (rc != VH_OK) return rc;
    VH_HIP(hipGetLastError());
    return VH_OK;
}

extern "C" int vh_integrate(vh_context *c, const float pose[16], const vh_float4 *verts, const vh_float4 *normals)
{
    (void)normals;
    if (!c || !pose || !verts) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    DeviceGuard guard(c->device);
    int rc = vh_set_pose(c, pose);
    if (rc == VH_OK) rc = vh_reset_mutexes(c);
    if (rc != VH_OK) return rc;
    if (c->fusedFrame && c->flattenVariant == kWalkMask) {
        // mask form: {claim || pure-stream walk that stores allocation masks}, then
        // {commit || consume the masks: frustum test, compaction, TSDF update}
        const uint32_t claimBlocks = (uint32_t)grid_for((size_t)c->fp.width * c->fp.height, 256);
        const uint32_t tiles = (uint32_t)grid_for(c->numEntries, kFlattenThreads * kEntriesPerLane);
        rc = launch(c, kPhaseFrameScanClaim, frame_mask_claim_kernel, dim3(claimBlocks + tiles), dim3(256), c->fp,
                    c->dp, reinterpret_cast<const float4 *>(verts), (uint32_t)c->numEntries, claimBlocks,
                    c->fusedParity);
        if (rc != VH_OK) return rc;
        const uint32_t commitBlocks = (uint32_t)c->commitBlocks;
        const uint32_t chunks = (uint32_t)grid_for(c->numEntries, kMaskChunkEntries);
        rc = launch(c, kPhaseFrameCommitIntegrate, frame_commit_consume_kernel, dim3(commitBlocks + chunks), dim3(256),
                    c->fp, c->dp, reinterpret_cast<const float4 *>(verts), (uint32_t)c->numEntries, commitBlocks,
                    c->fusedParity);
        if (rc != VH_OK) return rc;
        c->occupiedCounter = kScanCount + c->fusedParity;   // this frame's slot counter = occupied count
        c->fusedParity ^= 1;
        c->compactArmed = false;
    } else if (c->fusedFrame) {
        // two launches: {claim || table walk}, then {commit + integrate}; see vh_kernels.hip
        c->occupiedCounter = kCompactCount;
        const uint32_t claimBlocks = (uint32_t)grid_for((size_t)c->fp.width * c->fp.height, 256);
        const uint32_t scanBlocks = walk_blocks(c);
        if (c->flattenVariant == kWalkIndexed)
            rc = launch(c, kPhaseFrameScanClaim, frame_scan_claim_kernel<kWalkIndexed>,
                        dim3(claimBlocks + scanBlocks), dim3(256), c->fp, c->dp,
                        reinterpret_cast<const float4 *>(verts), (uint32_t)c->numEntries, claimBlocks,
                        c->fusedParity);
        else if (c->flattenVariant == kWalkPersistent)
            rc = launch(c, kPhaseFrameScanClaim, frame_scan_claim_kernel<kWalkPersistent>,
                        dim3(claimBlocks + scanBlocks), dim3(256), c->fp, c->dp,
                        reinterpret_cast<const float4 *>(verts), (uint32_t)c->numEntries, claimBlocks,
                        c->fusedParity);
        else if (c->flattenVariant == kWalkWide)
            rc = launch(c, kPhaseFrameScanClaim, frame_scan_claim_kernel<kWalkWide>, dim3(claimBlocks + scanBlocks),
                        dim3(256), c->fp, c->dp, reinterpret_cast<const float4 *>(verts), (uint32_t)c->numEntries,
                        claimBlocks, c->fusedParity);
        else if (c->flattenVariant == kWalkStridedBallot)
            rc = launch(c, kPhaseFrameScanClaim, frame_scan_claim_kernel<kWalkStridedBallot>,
                        dim3(claimBlocks + scanBlocks), dim3(256), c->fp, c->dp,
                        reinterpret_cast<const float4 *>(verts), (uint32_t)c->numEntries, claimBlocks,
                        c->fusedParity);
        else
            rc = launch(c, kPhaseFrameScanClaim, frame_scan_claim_kernel<kWalkStrided>,
                        dim3(claimBlocks + scanBlocks), dim3(256), c->fp, c->dp,
                        reinterpret_cast<const float4 *>(verts), (uint32_t)c->numEntries, claimBlocks,
                        c->fusedParity);
        if (rc != VH_OK) return rc;
        const uint32_t commitBlocks = (uint32_t)c->commitBlocks;
        rc = launch(c, kPhaseFrameCommitIntegrate, frame_commit_integrate_kernel,
                    dim3(commitBlocks + (uint32_t)c->integrateGrid), dim3(256), c->fp, c->dp,
                    reinterpret_cast<const float4 *>(verts), commitBlocks, c->fusedParity);
        if (rc != VH_OK) return rc;
        c->fusedParity ^= 1;       // this frame cleared the other counter set for the next one
        c->compactArmed = false;
    } else {
        // alloc_commit re-arms the compact counter, so no memset node is needed here
        c->occupiedCounter = kCompactCount;
        if ((rc = launch_alloc(c, verts)) != VH_OK) return rc;
        c->compactArmed = false;
        if ((rc = launch_flatten(c)) != VH_OK) return rc;
        if ((rc = launch_integrate(c, verts)) != VH_OK) return rc;
    }
    if (c->profiling) c->profiledFrames += 1;
    VH_HIP(hipGetLastError());
    return VH_OK;
}

// The frame straight from the uint16 sensor image: preProcess's vertex computation happens inside
// the claim half, the TSDF update reads the image.  Equals vh_preprocess + vh_integrate.
extern "C" int vh_integrate_depth(vh_context *c, const float pose[16], const uint16_t *d_depth, const float k_inv[9])
{
    if (!c || !pose || !d_depth || !k_inv) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    DeviceGuard guard(c->device);
    int rc = vh_set_pose(c, pose);
    if (rc == VH_OK) rc = vh_reset_mutexes(c);
    if (rc != VH_OK) return rc;
    SensorImage in;
    in.depth = d_depth;
    std::memcpy(in.k, k_inv, sizeof in.k);
    in.unit = 5000.0f;                                                   // CameraTrackingUtils.cu:64
    c->occupiedCounter = kCompactCount;
    const uint32_t claimBlocks = (uint32_t)grid_for((size_t)c->fp.width * c->fp.height, 256);
    const uint32_t scanBlocks = (uint32_t)grid_for(c->numEntries, kFlattenThreads * kEntriesPerLane);
    rc = launch(c, kPhaseFrameScanClaim, frame_scan_claim_sensor_kernel, dim3(claimBlocks + scanBlocks), dim3(256), c->fp,
                c->dp, in, (uint32_t)c->numEntries, claimBlocks, c->fusedParity);
    if (rc != VH_OK) return rc;
    const uint32_t commitBlocks = (uint32_t)c->commitBlocks;
    rc = launch(c, kPhaseFrameCommitIntegrate, frame_commit_integrate_sensor_kernel,
                dim3(commitBlocks + (uint32_t)c->integrateGrid), dim3(256), c->fp, c->dp, in, commitBlocks, c->fusedParity);
    if (rc != VH_OK) return rc;
    c->fusedParity ^= 1;
    c->compactArmed = false;
    if (c->profiling) c->profiledFrames += 1;
    VH_HIP(hipGetLastError());
    return VH_OK;
}

extern "C" int vh_raycast(vh_context *c, const float pose[16], float t_min, float t_max, float *d_depth_out)
{
    if (!c || !pose || !d_depth_out) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    if (!(t_max > t_min)) return fail(VH_ERR_INVALID_ARGUMENT, "t_max must exceed t_min");
    DeviceGuard guard(c->device);
    FrameParams fp = c->fp;
    std::memcpy(fp.T, pose, sizeof fp.T);
    const float q = (t_max - t_min) / fp.voxelSize;
    int nsteps = (q >= 2147483648.0f) ? 0x7fffffff : (int)q;
    nsteps += 1;
    dim3 grid((fp.width + 15) / 16, (fp.height + 15) / 16);
    DevPtrs dp = c->dp;
    if (c->viewBlocks) dp.blocks = const_cast<Voxel *>(c->viewBlocks);     // view table: voxels live in the records
    const int rc = c->raycastPatch
                       ? launch(c, kPhaseRaycast, raycast_kernel<1>, grid, dim3(256), fp, dp, c->rc_fx, c->rc_fy,
                                c->rc_cx, c->rc_cy, t_min, nsteps, d_depth_out, c->raycastXcd)
                       : launch(c, kPhaseRaycast, raycast_kernel<0>, grid, dim3(256), fp, dp, c->rc_fx, c->rc_fy,
                                c->rc_cx, c->rc_cy, t_min, nsteps, d_depth_out, c->raycastXcd);
    if (rc != VH_OK) return rc;
    VH_HIP(hipGetLastError());
    return VH_OK;
}

// Block silhouettes: SDFRenderer::drawToFrontAndBack, SDFRenderer.cpp:165-208.
extern "C" int vh_render_blocks(vh_context *c, const float pose[16], float t_min, float t_max, float *d_front,
                                float *d_back)
{
    if (!c || !pose || !d_front || !d_back) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    if (!(t_max > t_min) || !(t_min >= 0.0f)) return fail(VH_ERR_INVALID_ARGUMENT, "need 0 <= t_min < t_max");
    DeviceGuard guard(c->device);
    BlockView bv;
    float inv[16];
    invert4x4(pose, inv);
    std::memcpy(bv.T, pose, sizeof bv.T);
    std::memcpy(bv.Tinv, inv, sizeof bv.Tinv);
    bv.fx = c->rc_fx; bv.fy = c->rc_fy; bv.cx = c->rc_cx; bv.cy = c->rc_cy;
    bv.tMin = t_min;
    bv.tMax = t_max;
    const int32_t npix = c->fp.width * c->fp.height;
    // list of the allocated entries: room for every entry of the table, allocated on first use (synchronises once)
    if (!c->blockList) {
        VH_HIP(hipStreamSynchronize(c->stream));
        VH_HIP(hipMalloc((void **)&c->blockList, sizeof(int32_t) * (c->numEntries + 4)));
    }
    int32_t *listCount = c->blockList;
    int32_t *list = listCount + 4;
    const int32_t capacity = (int32_t)c->numEntries;
    uint32_t *front = reinterpret_cast<uint32_t *>(d_front), *back = reinterpret_cast<uint32_t *>(d_back);
    int rc = launch(c, kPhaseRaycastBounds, blocks_init_kernel, dim3((unsigned)grid_for((size_t)npix, 256)), dim3(256), front,
                    back, npix, listCount);
    const uint32_t words = (c->ownedBuckets + 31u) / 32u;
    if (rc == VH_OK)
        rc = launch(c, kPhaseRaycastBounds, blocks_list_kernel, dim3((unsigned)grid_for(words, 256)), dim3(256), c->fp, c->dp,
                    list, capacity, listCount);
    if (rc == VH_OK)
        rc = launch(c, kPhaseRaycastBounds, blocks_raster_kernel, dim3(1024, 16), dim3(256), c->fp, c->dp, bv,
                    (const int32_t *)list, capacity, (const int32_t *)listCount, front, back);
    if (rc == VH_OK)
        rc = launch(c, kPhaseRaycastBounds, blocks_finish_kernel, dim3((unsigned)grid_for((size_t)npix, 256)), dim3(256), front,
                    npix);
    if (rc != VH_OK) return rc;
    VH_HIP(hipGetLastError());
    return VH_OK;
}
