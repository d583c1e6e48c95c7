// vh_api_shard.hip -- C-ABI, bucket-range shards: view export / import for the raycast over shards, key bins, camera packets, batched frames.
// Included by vh_api.hip (same translation unit: shares fail(), VH_HIP, DeviceGuard, launch()).

// ---------------------------------------------------------------------------
// raycast over shards: export of the blocks a view can touch, import into a view table
// ---------------------------------------------------------------------------
// oracle: vho_view_frustum (same operations in the same order)
static void make_view_frustum(const vh_context *c, const float pose[16], float t_min, float t_max, float f[22])
{
    float inv[16];
    invert4x4(pose, inv);
    std::memcpy(f, inv, 12 * sizeof(float));
    const float r = 7.0f * c->fp.voxelSize;
    const float a[4] = {(0.0f - c->rc_cx) / c->rc_fx, ((float)(c->fp.width - 1) - c->rc_cx) / c->rc_fx,
                        (0.0f - c->rc_cy) / c->rc_fy, ((float)(c->fp.height - 1) - c->rc_cy) / c->rc_fy};
    for (int i = 0; i < 4; ++i) {
        f[12 + i] = a[i];
        f[16 + i] = -(r * sqrtf(1.0f + a[i] * a[i]));
    }
    f[20] = t_min - r;
    f[21] = t_max + r;
}

extern "C" int vh_export_views(vh_context *c, const float *poses, int32_t n_views, float t_min, float t_max,
                               vh_view_record *d_records, int32_t capacity, int32_t *d_counts)
{
    if (!c || !poses || !d_records || !d_counts) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    if (n_views < 1 || n_views > VH_MAX_CAMERAS) return fail(VH_ERR_INVALID_ARGUMENT, "1..VH_MAX_CAMERAS views");
    if (capacity < 1) return fail(VH_ERR_INVALID_ARGUMENT, "capacity must be positive");
    if (!(t_max > t_min)) return fail(VH_ERR_INVALID_ARGUMENT, "t_max must exceed t_min");
    if (c->viewBlocks) return fail(VH_ERR_INVALID_ARGUMENT, "a view table has no voxels of its own to export");
    DeviceGuard guard(c->device);
    { const int frc = flush_pending(c); if (frc != VH_OK) return frc; }
    const size_t need = (size_t)n_views * (size_t)capacity;
    if (c->viewListsSize < need) {                         // first call (or a larger one): synchronises
        VH_HIP(hipStreamSynchronize(c->stream));
        if (c->viewLists) (void)hipFree(c->viewLists);
        c->viewLists = nullptr;
        c->viewListsSize = 0;
        VH_HIP(hipMalloc((void **)&c->viewLists, need * sizeof(int32_t)));
        c->viewListsSize = need;
    }
    VH_HIP(hipMemsetAsync(d_counts, 0, sizeof(int32_t) * (size_t)n_views, c->stream));
    const uint32_t tiles = (uint32_t)((c->numEntries + kFlattenThreads * kEntriesPerLane - 1) /
                                      (kFlattenThreads * kEntriesPerLane));
    for (int32_t base = 0; base < n_views; base += kMaxViewsPerLaunch) {
        const int32_t n = std::min<int32_t>(kMaxViewsPerLaunch, n_views - base);
        ViewSet vs;
        std::memset(&vs, 0, sizeof vs);
        for (int32_t v = 0; v < n; ++v) make_view_frustum(c, poses + 16 * (size_t)(base + v), t_min, t_max, vs.v[v].f);
        const int rc = launch(c, kPhaseViewExport, view_select_kernel, dim3(tiles), dim3(kFlattenThreads), c->fp, c->dp,
                              (uint32_t)c->numEntries, vs, n, c->viewLists + (size_t)base * capacity, capacity,
                              d_counts + base);
        if (rc != VH_OK) return rc;
    }
    const int rc = launch(c, kPhaseViewExport, view_pack_kernel, dim3((unsigned)std::min<int32_t>(capacity, 2048), n_views),
                          dim3(256), c->dp, (const int32_t *)c->viewLists, (const int32_t *)d_counts, capacity,
                          reinterpret_cast<uint8_t *>(d_records), 0);
    if (rc != VH_OK) return rc;
    VH_HIP(hipGetLastError());
    return VH_OK;
}

extern "C" int vh_import_view(vh_context *c, const vh_view_record *d_records, int32_t count)
{
    if (!c || (!d_records && count > 0)) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    if (count < 0 || (size_t)count > c->numEntries || (uint64_t)count * kViewRecordVoxels + 514ull > 0x7fffffffull)
        return fail(VH_ERR_INVALID_ARGUMENT, "bad record count");
    if (c->fp.bucketLo != 0 || c->fp.bucketHi != c->fp.numBuckets)
        return fail(VH_ERR_INVALID_ARGUMENT, "a view table is unsharded");
    if (c->epochTotal != 0) return fail(VH_ERR_INVALID_ARGUMENT, "this context has integrated frames: use a dedicated view context");
    DeviceGuard guard(c->device);             // (before ensure_candidates: it allocates and synchronises on c's device)
    if ((size_t)count > c->candAllocated && (c->fp.flags & kFlagOverflow)) {
        const int rc = ensure_candidates(c, (size_t)count);
        if (rc != VH_OK) return rc;
    }
    { const int frc = flush_pending(c); if (frc != VH_OK) return frc; }
    if (c->viewCount > 0) {
        const int rc = launch(c, kPhaseViewImport, view_clear_kernel, dim3((unsigned)grid_for((size_t)c->viewCount, 256)),
                              dim3(256), c->fp, c->dp, c->viewCount);
        if (rc != VH_OK) return rc;
    }
    VH_HIP(hipMemsetAsync(c->dp.bucketBits, 0, sizeof(uint32_t) * (((size_t)c->ownedBuckets + 31) / 32), c->stream));
    VH_HIP(hipMemsetAsync(c->dp.macroBits, 0, kMacroBits / 8, c->stream));
    c->viewBlocks = reinterpret_cast<const Voxel *>(d_records);
    c->viewCount = count;
    if (count > 0) {
        int rc = launch(c, kPhaseViewImport, view_import_kernel, dim3((unsigned)grid_for((size_t)count, 256)),
                        dim3(256), c->fp, c->dp, reinterpret_cast<const uint8_t *>(d_records), count,
                        (const int32_t *)nullptr, 0);
        if (rc == VH_OK && (c->fp.flags & kFlagOverflow))
            rc = launch(c, kPhaseViewImport, view_import_overflow_kernel, dim3(1), dim3(64), c->fp, c->dp,
                        reinterpret_cast<const uint8_t *>(d_records));
        if (rc != VH_OK) return rc;
    }
    VH_HIP(hipGetLastError());
    return VH_OK;
}

// The same round without a host synchronisation: view poses stay on the device (they arrive by
// all-gather), every view's records go to a fixed slot range [v * capacity, (v + 1) * capacity) so the
// exchange has equal, host-known sizes, and the importer reads the counts on the device.
extern "C" int vh_export_views_fixed(vh_context *c, const float *d_poses, int32_t n_views, float t_min, float t_max,
                                     vh_view_record *d_records, int32_t capacity, int32_t *d_counts)
{
    if (!c || !d_poses || !d_records || !d_counts) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    if (n_views < 1 || n_views > kMaxViewsPerLaunch) return fail(VH_ERR_INVALID_ARGUMENT, "1..16 views");
    if (capacity < 1) return fail(VH_ERR_INVALID_ARGUMENT, "capacity must be positive");
    if (!(t_max > t_min)) return fail(VH_ERR_INVALID_ARGUMENT, "t_max must exceed t_min");
    if (c->viewBlocks) return fail(VH_ERR_INVALID_ARGUMENT, "a view table has no voxels of its own to export");
    DeviceGuard guard(c->device);
    { const int frc = flush_pending(c); if (frc != VH_OK) return frc; }
    const size_t need = (size_t)n_views * (size_t)capacity;
    if (c->viewListsSize < need || !c->viewSet) {            // first call (or a larger one): synchronises
        VH_HIP(hipStreamSynchronize(c->stream));
        if (c->viewListsSize < need) {
            if (c->viewLists) (void)hipFree(c->viewLists);
            c->viewLists = nullptr;
            c->viewListsSize = 0;
            VH_HIP(hipMalloc((void **)&c->viewLists, need * sizeof(int32_t)));
            c->viewListsSize = need;
        }
        if (!c->viewSet) VH_HIP(hipMalloc((void **)&c->viewSet, sizeof(ViewSet)));
    }
    VH_HIP(hipMemsetAsync(d_counts, 0, sizeof(int32_t) * (size_t)n_views, c->stream));
    const ViewCamera cam{c->rc_fx, c->rc_fy, c->rc_cx, c->rc_cy, t_min, t_max, c->fp.width, c->fp.height};
    int rc = launch(c, kPhaseViewExport, view_frustum_kernel, dim3(1), dim3(64), d_poses, n_views, cam, c->fp.voxelSize,
                    reinterpret_cast<ViewSet *>(c->viewSet));
    const uint32_t tiles = (uint32_t)((c->numEntries + kFlattenThreads * kEntriesPerLane - 1) /
                                      (kFlattenThreads * kEntriesPerLane));
    if (rc == VH_OK)
        rc = launch(c, kPhaseViewExport, view_select_mem_kernel, dim3(tiles), dim3(kFlattenThreads), c->fp, c->dp,
                    (uint32_t)c->numEntries, (const ViewSet *)c->viewSet, n_views, c->viewLists, capacity, d_counts);
    if (rc == VH_OK)
        rc = launch(c, kPhaseViewExport, view_pack_kernel, dim3((unsigned)std::min<int32_t>(capacity, 2048), n_views),
                    dim3(256), c->dp, (const int32_t *)c->viewLists, (const int32_t *)d_counts, capacity,
                    reinterpret_cast<uint8_t *>(d_records), 1);
    if (rc != VH_OK) return rc;
    VH_HIP(hipGetLastError());
    return VH_OK;
}

extern "C" int vh_import_views(vh_context *c, const vh_view_record *d_records, int32_t num_sources, int32_t capacity,
                               const int32_t *d_counts)
{
    if (!c || !d_records || num_sources < 1 || capacity < 1) return fail(VH_ERR_INVALID_ARGUMENT, "bad argument");
    const size_t slots = (size_t)num_sources * (size_t)capacity;
    if (slots > c->numEntries || (uint64_t)slots * kViewRecordVoxels + 514ull > 0x7fffffffull)
        return fail(VH_ERR_INVALID_ARGUMENT, "too many record slots for this view table");
    if (c->fp.bucketLo != 0 || c->fp.bucketHi != c->fp.numBuckets)
        return fail(VH_ERR_INVALID_ARGUMENT, "a view table is unsharded");
    if (c->epochTotal != 0) return fail(VH_ERR_INVALID_ARGUMENT, "this context has integrated frames: use a dedicated view context");
    DeviceGuard guard(c->device);
    if ((c->fp.flags & kFlagOverflow) && slots > c->candAllocated) {
        const int rc = ensure_candidates(c, slots);
        if (rc != VH_OK) return rc;
    }
    if (c->viewCount > 0) {
        const int rc = launch(c, kPhaseViewImport, view_clear_kernel, dim3((unsigned)grid_for((size_t)c->viewCount, 256)),
                              dim3(256), c->fp, c->dp, c->viewCount);
        if (rc != VH_OK) return rc;
    }
    VH_HIP(hipMemsetAsync(c->dp.bucketBits, 0, sizeof(uint32_t) * (((size_t)c->ownedBuckets + 31) / 32), c->stream));
    VH_HIP(hipMemsetAsync(c->dp.macroBits, 0, kMacroBits / 8, c->stream));
    c->viewBlocks = reinterpret_cast<const Voxel *>(d_records);
    c->viewCount = (int32_t)slots;
    int rc = launch(c, kPhaseViewImport, view_import_kernel, dim3((unsigned)grid_for(slots, 256)), dim3(256), c->fp, c->dp,
                    reinterpret_cast<const uint8_t *>(d_records), (int32_t)slots, d_counts, capacity);
    if (rc == VH_OK && (c->fp.flags & kFlagOverflow))
        rc = launch(c, kPhaseViewImport, view_import_overflow_kernel, dim3(1), dim3(64), c->fp, c->dp,
                    reinterpret_cast<const uint8_t *>(d_records));
    if (rc != VH_OK) return rc;
    VH_HIP(hipGetLastError());
    return VH_OK;
}

// ---------------------------------------------------------------------------
// sharding
// ---------------------------------------------------------------------------
// 4-byte units of one camera packet in the context's packet format
static size_t packet_units(const vh_context *c)
{
    const size_t npix = (size_t)c->fp.width * c->fp.height;
    return c->packetFormat == VH_PACKET_U16 ? (size_t)kPacketHeaderU16 + (npix + 1) / 2 : (size_t)kPacketHeader + npix;
}

extern "C" int vh_generate_keys(vh_context *c, const vh_float4 *verts, uint32_t camera_id, int32_t num_shards,
                                int32_t *d_bins, int32_t capacity, int32_t bin_stride, float *d_packet)
{
    if (bin_stride == 0) bin_stride = capacity;
    if (!c || !verts || !d_bins || num_shards <= 0 || capacity < 2 || bin_stride < capacity ||
        camera_id >= VH_MAX_CAMERAS || num_shards > VH_MAX_CAMERAS)
        return fail(VH_ERR_INVALID_ARGUMENT, "bad argument");
    DeviceGuard guard(c->device);
    prepare_bins_kernel<<<1, 64, 0, c->stream>>>(reinterpret_cast<int4 *>(d_bins), num_shards, bin_stride, 1, 0);
    generate_keys_kernel<<<grid_for(host_num_tiles(c), kGenTiles), kGenThreads, 0, c->stream>>>(
        c->fp, reinterpret_cast<const float4 *>(verts), num_shards, reinterpret_cast<int4 *>(d_bins), capacity,
        bin_stride, d_packet ? d_packet + kPacketHeader : nullptr, camera_id << kRankCameraShift);
    VH_HIP(hipGetLastError());
    return VH_OK;
}

// Per-batch bins: behind every generation launch but the last, the count so far is written into the bin headers as a mark, so
// that the launch of a frame reads only the records of its own generation launch (claim_bin_slice).  At most two marks fit the
// header: batches of more than three generation launches go without (their frames scan the whole bin, as before).
static void mark_bins(vh_context *c, bool perBatch, int batch, int b0, int4 *bins, int32_t numShards, int32_t binStride)
{
    const int per = c->genFramesPerLaunch, launches = (batch + per - 1) / per, g = b0 / per;
    if (!perBatch || launches < 2 || launches > 3 || g >= launches - 1 || per > 255) return;
    bin_mark_kernel<<<1, 64, 0, c->stream>>>(bins, numShards, binStride, g, per, launches - 1);
}

// `batch` frames of one camera in one call: one launch zeroes all bin headers, then one launch per
// kGenBatch frames (blockIdx.y = frame, poses and vertex-map pointers in the kernel arguments).
extern "C" int vh_generate_keys_batch(vh_context *c, int32_t batch, const float *poses,
                                      const vh_float4 *const *d_verts, uint32_t camera_id, int32_t num_shards,
                                      int32_t *d_bins, int32_t capacity, int32_t bin_stride, int32_t frame_stride,
                                      float *d_packets, size_t packet_frame_stride)
{
    if (!c || !poses || !d_verts || !d_bins || batch <= 0 || num_shards <= 0 || num_shards > VH_MAX_CAMERAS ||
        capacity < 2 || camera_id >= VH_MAX_CAMERAS)
        return fail(VH_ERR_INVALID_ARGUMENT, "bad argument");
    const bool perBatch = frame_stride == VH_BIN_PER_BATCH;            // one bin per owner for all frames of the batch
    if (perBatch && batch > (int32_t)VH_MAX_CAMERAS) return fail(VH_ERR_INVALID_ARGUMENT, "a per-batch bin holds at most 32 frames");
    if (frame_stride == 0) frame_stride = capacity;
    if (bin_stride == 0) bin_stride = perBatch ? capacity : batch * frame_stride;
    const size_t dense = (size_t)kPacketHeader + (size_t)c->fp.width * c->fp.height;
    if (packet_frame_stride == 0) packet_frame_stride = dense;
    if ((!perBatch && (frame_stride < capacity || bin_stride < batch * frame_stride)) || (perBatch && bin_stride < capacity) || (d_packets && packet_frame_stride < dense))
        return fail(VH_ERR_INVALID_ARGUMENT, "bad stride");
    DeviceGuard guard(c->device);
    if (perBatch)
        prepare_bins_kernel<<<1, 64, 0, c->stream>>>(reinterpret_cast<int4 *>(d_bins), num_shards, bin_stride, 1, 0);
    else
        prepare_bins_kernel<<<grid_for((size_t)num_shards * batch, 256), 256, 0, c->stream>>>(
            reinterpret_cast<int4 *>(d_bins), num_shards, bin_stride, batch, frame_stride);
    for (int b0 = 0; b0 < batch; b0 += c->genFramesPerLaunch) {
        const int n = std::min<int>(c->genFramesPerLaunch, batch - b0);
        GenFrames fr;
        std::memset(&fr, 0, sizeof fr);
        for (int j = 0; j < n; ++j) {
            const int rc = vh_set_pose(c, poses + 16 * (size_t)(b0 + j));        // pose + cofactor inverse
            if (rc != VH_OK) return rc;
            if (!d_verts[b0 + j]) return fail(VH_ERR_INVALID_ARGUMENT, "null vertex map");
            std::memcpy(fr.T[j], c->fp.T, sizeof fr.T[j]);
            std::memcpy(fr.Tinv[j], c->fp.Tinv, sizeof fr.Tinv[j]);
            fr.verts[j] = reinterpret_cast<const float4 *>(d_verts[b0 + j]);
        }
        if (num_shards >= 4)
            generate_keys_batch_kernel<kGenThreads / 2><<<dim3((unsigned)grid_for(host_num_tiles(c), kGenTiles / 2), (unsigned)n), kGenThreads / 2, 0, c->stream>>>(
            c->fp, fr, num_shards, perBatch ? reinterpret_cast<int4 *>(d_bins) : reinterpret_cast<int4 *>(d_bins) + (size_t)frame_stride * b0, capacity, bin_stride,
            perBatch ? -1 : frame_stride, d_packets ? d_packets + packet_frame_stride * (size_t)b0 : nullptr, packet_frame_stride,
            perBatch ? (uint32_t)b0 << kRankCameraShift : camera_id << kRankCameraShift);
        else
            generate_keys_batch_kernel<kGenThreads><<<dim3((unsigned)grid_for(host_num_tiles(c), kGenTiles), (unsigned)n), kGenThreads, 0, c->stream>>>(
            c->fp, fr, num_shards, perBatch ? reinterpret_cast<int4 *>(d_bins) : reinterpret_cast<int4 *>(d_bins) + (size_t)frame_stride * b0, capacity, bin_stride,
            perBatch ? -1 : frame_stride, d_packets ? d_packets + packet_frame_stride * (size_t)b0 : nullptr, packet_frame_stride,
            perBatch ? (uint32_t)b0 << kRankCameraShift : camera_id << kRankCameraShift);
        mark_bins(c, perBatch, batch, b0, reinterpret_cast<int4 *>(d_bins), num_shards, bin_stride);
    }
    VH_HIP(hipGetLastError());
    return VH_OK;
}

// Keys and sensor-depth packets of `batch` frames of this camera from the uint16 images alone: one
// launch per kGenBatch frames (no vertex map, no separate packet pass).
template <int kThreads, int kGroups>
static void launch_gen_sensor(vh_context *c, int n, const GenSensorFrames &fr, int32_t numShards, int4 *bins, int32_t capacity,
                              int32_t binStride, int32_t frameStride, float *packets, size_t packetFrameStride, uint32_t rankBase)
{
    const unsigned perGroup = (unsigned)(kThreads / 256) * (unsigned)(kGroups > 0 ? kGroups : 1);
    generate_keys_sensor_batch_kernel<kThreads, kGroups><<<dim3((unsigned)grid_for(host_num_tiles(c), perGroup), (unsigned)n), kThreads, 0, c->stream>>>(
        c->fp, fr, numShards, bins, capacity, binStride, frameStride, packets, packetFrameStride, rankBase);
}

#ifndef VH_GEN_THREADS_ONE
#define VH_GEN_THREADS_ONE (kGenThreads / 2)
#endif
#ifndef VH_GEN_GROUPS_ONE
#define VH_GEN_GROUPS_ONE 8        // pixel groups per key-generation workgroup, fewer than four owners (R = 1, batch of 8: 35.7 us;
                                   // 1024 lanes x 2 / 4 / 8 groups 38.2 / 38.7 / 41.5, 512 x 4 39.0, 256 x 8 / 16 37.1 / 43.6)
#endif
#ifndef VH_GEN_GROUPS_MANY
#define VH_GEN_GROUPS_MANY 2       // ... four owners or more (512-lane workgroups; R = 8: 23.6 us with 1, 2 or 4 groups -- most of the
                                   // step from 28.1 is the band code this path does not carry)
#endif
extern "C" int vh_generate_keys_depth_batch(vh_context *c, int32_t batch, const float *poses,
                                            const uint16_t *const *d_depth, const float k_inv[9], uint32_t camera_id,
                                            int32_t num_shards, int32_t *d_bins, int32_t capacity, int32_t bin_stride,
                                            int32_t frame_stride, float *d_packets, size_t packet_frame_stride)
{
    if (!c || !poses || !d_depth || !k_inv || !d_bins || batch <= 0 || num_shards <= 0 ||
        num_shards > VH_MAX_CAMERAS || capacity < 2 || camera_id >= VH_MAX_CAMERAS)
        return fail(VH_ERR_INVALID_ARGUMENT, "bad argument");
    const size_t npix = (size_t)c->fp.width * c->fp.height;
    if (d_packets && npix % 2) return fail(VH_ERR_INVALID_ARGUMENT, "sensor-depth packets need an even number of pixels");
    const bool perBatch = frame_stride == VH_BIN_PER_BATCH;            // one bin per owner for all frames of the batch
    if (perBatch && batch > (int32_t)VH_MAX_CAMERAS) return fail(VH_ERR_INVALID_ARGUMENT, "a per-batch bin holds at most 32 frames");
    if (frame_stride == 0) frame_stride = capacity;
    if (bin_stride == 0) bin_stride = perBatch ? capacity : batch * frame_stride;
    const size_t dense = (size_t)kPacketHeaderU16 + npix / 2;
    if (packet_frame_stride == 0) packet_frame_stride = dense;
    if ((!perBatch && (frame_stride < capacity || bin_stride < batch * frame_stride)) || (perBatch && bin_stride < capacity) || (d_packets && packet_frame_stride < dense))
        return fail(VH_ERR_INVALID_ARGUMENT, "bad stride");
    DeviceGuard guard(c->device);
    if (perBatch)
        prepare_bins_kernel<<<1, 64, 0, c->stream>>>(reinterpret_cast<int4 *>(d_bins), num_shards, bin_stride, 1, 0);
    else
        prepare_bins_kernel<<<grid_for((size_t)num_shards * batch, 256), 256, 0, c->stream>>>(
            reinterpret_cast<int4 *>(d_bins), num_shards, bin_stride, batch, frame_stride);
    for (int b0 = 0; b0 < batch; b0 += c->genFramesPerLaunch) {
        const int n = std::min<int>(c->genFramesPerLaunch, batch - b0);
        GenSensorFrames fr;
        std::memset(&fr, 0, sizeof fr);
        std::memcpy(fr.k, k_inv, sizeof fr.k);
        fr.unit = 5000.0f;                                               // CameraTrackingUtils.cu:64
        for (int j = 0; j < n; ++j) {
            const int rc = vh_set_pose(c, poses + 16 * (size_t)(b0 + j));
            if (rc != VH_OK) return rc;
            if (!d_depth[b0 + j]) return fail(VH_ERR_INVALID_ARGUMENT, "null depth image");
            std::memcpy(fr.T[j], c->fp.T, sizeof fr.T[j]);
            std::memcpy(fr.Tinv[j], c->fp.Tinv, sizeof fr.Tinv[j]);
            fr.depth[j] = d_depth[b0 + j];
        }
        // no band: several pixel groups per workgroup, one atomic per owner for all of them (vh_alloc.hip)
        int4 *bins0 = perBatch ? reinterpret_cast<int4 *>(d_bins) : reinterpret_cast<int4 *>(d_bins) + (size_t)frame_stride * b0;
        float *pk0 = d_packets ? d_packets + packet_frame_stride * (size_t)b0 : nullptr;
        const uint32_t rank0 = perBatch ? (uint32_t)b0 << kRankCameraShift : camera_id << kRankCameraShift;
        const int fs = perBatch ? -1 : frame_stride;
        const bool band = c->fp.allocBand > 0.0f, many = num_shards >= 4;
        if (band && many) launch_gen_sensor<kGenThreads / 2, 0>(c, n, fr, num_shards, bins0, capacity, bin_stride, fs, pk0, packet_frame_stride, rank0);
        else if (band) launch_gen_sensor<kGenThreads, 0>(c, n, fr, num_shards, bins0, capacity, bin_stride, fs, pk0, packet_frame_stride, rank0);
        else if (many) launch_gen_sensor<kGenThreads / 2, VH_GEN_GROUPS_MANY>(c, n, fr, num_shards, bins0, capacity, bin_stride, fs, pk0, packet_frame_stride, rank0);
        else launch_gen_sensor<VH_GEN_THREADS_ONE, VH_GEN_GROUPS_ONE>(c, n, fr, num_shards, bins0, capacity, bin_stride, fs, pk0, packet_frame_stride, rank0);
        mark_bins(c, perBatch, batch, b0, reinterpret_cast<int4 *>(d_bins), num_shards, bin_stride);
    }
    VH_HIP(hipGetLastError());
    return VH_OK;
}

// Sensor-depth packets (VH_PACKET_U16) of `batch` frames of this camera: pose + inverse + K_inv row 2 +
// depth unit, then the uint16 image as it is.  The key generation for the same frames is
// vh_generate_keys_batch with d_packets = NULL.
extern "C" int vh_write_packets_u16_batch(vh_context *c, int32_t batch, const float *poses,
                                          const uint16_t *const *d_depth, const float k_inv[9], float *d_packets,
                                          size_t packet_frame_stride)
{
    if (!c || !poses || !d_depth || !k_inv || !d_packets || batch <= 0)
        return fail(VH_ERR_INVALID_ARGUMENT, "bad argument");
    const size_t npix = (size_t)c->fp.width * c->fp.height;
    if (npix % 2) return fail(VH_ERR_INVALID_ARGUMENT, "sensor-depth packets need an even number of pixels");
    const size_t dense = (size_t)kPacketHeaderU16 + npix / 2;
    if (packet_frame_stride == 0) packet_frame_stride = dense;
    if (packet_frame_stride < dense) return fail(VH_ERR_INVALID_ARGUMENT, "bad stride");
    DeviceGuard guard(c->device);
    for (int b0 = 0; b0 < batch; b0 += kGenBatch) {
        const int n = std::min<int>(kGenBatch, batch - b0);
        SensorFrames fr;
        std::memset(&fr, 0, sizeof fr);
        fr.k6 = k_inv[6]; fr.k7 = k_inv[7]; fr.k8 = k_inv[8];
        fr.unit = 5000.0f;                                               // CameraTrackingUtils.cu:64
        for (int j = 0; j < n; ++j) {
            const int rc = vh_set_pose(c, poses + 16 * (size_t)(b0 + j));
            if (rc != VH_OK) return rc;
            if (!d_depth[b0 + j]) return fail(VH_ERR_INVALID_ARGUMENT, "null depth image");
            std::memcpy(fr.T[j], c->fp.T, sizeof fr.T[j]);
            std::memcpy(fr.Tinv[j], c->fp.Tinv, sizeof fr.Tinv[j]);
            fr.depth[j] = d_depth[b0 + j];
        }
        write_packets_u16_kernel<<<dim3(64, (unsigned)n), 256, 0, c->stream>>>(
            fr, (int32_t)npix, d_packets + packet_frame_stride * (size_t)b0, packet_frame_stride);
    }
    VH_HIP(hipGetLastError());
    return VH_OK;
}

// One launch of the multi-camera pipeline: {claim || walk} of frame b of batch `mb` (null: none) and {commit + TSDF
// update} of the frame that is pending (c->multiPend), if any.
struct MultiBatch {
    const int4 *bins;
    const float *packets;
    int32_t numBins, capacity, binStride, frameStride, numCams;
    size_t packetStride, packetFrameStride;
    bool perBatch;           // the bins hold the whole batch (frameStride 0): a launch claims the records of its frame
};

static int launch_multi_pipelined(vh_context *c, const MultiBatch *mb, int b, const GenJob *job = nullptr)
{
    MultiPending &mp = c->multiPend;
    const bool doNew = mb != nullptr, hasOld = mp.active;
    if (!doNew && !hasOld) return VH_OK;
    int rc;
    if (doNew && (rc = vh_reset_mutexes(c)) != VH_OK) return rc;
    uint32_t *maskOf[2] = {c->dp.compactMask, c->maskBuf2};
    const int oldParity = c->pipeParity, newParity = oldParity ^ 1;
    const int setOld = c->pipeSet, setNew = (setOld + 1) % 3;
    MultiPipeArgs a;
    std::memset(&a, 0, sizeof a);
    a.numEntries = (uint32_t)c->numEntries;
    a.numCams = doNew ? mb->numCams : mp.numCams;
    a.packetStride = doNew ? mb->packetStride : mp.packetStride;
    if (doNew) {
        uint32_t parts = (uint32_t)grid_for((size_t)mb->capacity, 256 * 4);
        if (parts < 1) parts = 1;
        a.claimBlocks = (uint32_t)mb->numBins * parts;
        // entries per lane of the multi-camera walk by the shard's size: 4 from 32 MB of entries on, else 8
        a.walkShort = c->numEntries * sizeof(VoxelEntry) >= ((size_t)32 << 20) ? 1u : 0u;
        a.walkBlocks = (uint32_t)grid_for(c->numEntries, kFlattenThreads * (a.walkShort ? kEntriesPerLaneShort : kEntriesPerLane));
        // the walk-free multi-camera frame (flatten_variant 4; not with the overflow list: holes and chains take the reference's walk)
        a.walkIndexed = (c->flattenVariant == kWalkIndexed && !(c->fp.flags & kFlagOverflow)) ? 1u : 0u;
        if (a.walkIndexed) a.walkBlocks = (uint32_t)grid_for(((size_t)c->ownedBuckets + 31) / 32, kFlattenThreads * kIndexWords);
        a.partsPerBin = parts; a.numBins = (uint32_t)mb->numBins;
        a.capacity = mb->capacity; a.binStride = mb->binStride;
        a.binsNew = mb->bins + (size_t)mb->frameStride * b;
        a.binFrame = mb->perBatch ? b : -1;
        a.packetsNew = mb->packets + mb->packetFrameStride * b;
    }
    a.commitBlocks = hasOld ? (uint32_t)c->commitBlocks : 0u;
    a.integrateBlocks = hasOld ? (uint32_t)c->pipeIntegrateGrid : 0u;
    a.setNew = kPipeSetStride * setNew; a.setOld = kPipeSetStride * setOld; a.setClear = kPipeSetStride * ((setNew + 1) % 3);
    a.hasNew = doNew; a.hasOld = hasOld;
    a.claimSpan = claim_span(c, a.claimBlocks, a.walkBlocks);
    a.claimRatio = claim_ratio(a.claimBlocks, a.claimSpan);
    a.epochOld = mp.epochOld;
    a.doneTag = mp.doneTag;
    a.spinLimit = c->spinLimit ? c->spinLimit : kSpinLimitDefault;
    a.packetsOld = mp.packetsOld;
    DevPtrs dpNew = pipe_view(c, newParity);
    dpNew.compactMask = maskOf[newParity];
    const DevPtrs dpOld = pipe_view(c, oldParity);
    a.claimOld = dpOld.claim; a.candOld = dpOld.candidates; a.compactOld = dpOld.compact; a.maskOld = maskOf[oldParity];
    a.candCapacityOld = dpOld.candCapacity;
    const int format = doNew ? c->packetFormat : mp.packetFormat;
    const bool serial = (c->fp.flags & kFlagOverflow) != 0u;
    // the fused generation role (vh_dist): sensor frames, no overflow list (its launches may be serialised inside), no band
    const bool gen = job && job->blocks && doNew && format == VH_PACKET_U16 && !serial;
    if (gen) a.gen = *job;
    const dim3 grid(a.commitBlocks + a.integrateBlocks + (gen ? a.gen.blocks : 0u) + a.claimBlocks + a.walkBlocks);
    if (gen)
        rc = a.walkIndexed ? launch(c, kPhaseFramePipelined, frame_multi_pipelined_kernel<true, false, true, true>, grid, dim3(256), c->fp, dpNew, a)
                           : launch(c, kPhaseFramePipelined, frame_multi_pipelined_kernel<true, false, false, true>, grid, dim3(256), c->fp, dpNew, a);
    else if (a.walkIndexed)         // (never with the overflow list, hence never serialised)
        rc = format == VH_PACKET_U16 ? launch(c, kPhaseFramePipelined, frame_multi_pipelined_kernel<true, false, true>, grid, dim3(256), c->fp, dpNew, a)
                                     : launch(c, kPhaseFramePipelined, frame_multi_pipelined_kernel<false, false, true>, grid, dim3(256), c->fp, dpNew, a);
    else
    rc = format == VH_PACKET_U16
             ? (serial ? launch(c, kPhaseFramePipelined, frame_multi_pipelined_kernel<true, true>, grid, dim3(256), c->fp, dpNew, a)
                       : launch(c, kPhaseFramePipelined, frame_multi_pipelined_kernel<true, false>, grid, dim3(256), c->fp, dpNew, a))
             : (serial ? launch(c, kPhaseFramePipelined, frame_multi_pipelined_kernel<false, true>, grid, dim3(256), c->fp, dpNew, a)
                       : launch(c, kPhaseFramePipelined, frame_multi_pipelined_kernel<false, false>, grid, dim3(256), c->fp, dpNew, a));
    if (rc != VH_OK) return rc;
    if (serial && doNew && hasOld) c->serialQueued = true;         // (its claim / walk workgroups wait: check_spin_timeouts)
    if (doNew) {
        mp.active = true;
        mp.epochOld = c->fp.epoch;
        mp.doneTag = (int32_t)(c->epochTotal & 0x7fffffffu) | 0x40000000;
        mp.packetsOld = a.packetsNew;
        mp.packetStride = mb->packetStride;
        mp.numCams = mb->numCams;
        mp.packetFormat = c->packetFormat;
        c->pipeSet = setNew;
        c->pipeParity = newParity;
        c->dp.claim = dpNew.claim; c->dp.candidates = dpNew.candidates; c->dp.compact = dpNew.compact;
    } else {
        mp.active = false;
    }
    c->occupiedCounter = kCompactCount;
    c->compactArmed = false;
    c->foldA = -1;
    if (c->profiling && hasOld) c->profiledFrames += 1;
    return VH_OK;
}

// the pending multi-camera frame's deferred half in a launch of its own (flush_pending: every observer comes through it)
static int flush_multi_pending(vh_context *c)
{
    if (!c->multiPend.active) return VH_OK;
    const int rc = launch_multi_pipelined(c, nullptr, 0);
    if (rc != VH_OK) return rc;
    VH_HIP(hipGetLastError());
    return VH_OK;
}


// `batch` multi-camera frames applied one after the other, each as the fused pair of launches
// (new lock epoch; {claim bins || walk}; {commit + integrate}).
// Can the multi-camera frames of this context carry a fused generation role (launch_multi_pipelined: gen)?  What vh_dist asks
// before it hands vh_apply_frames_batch_gen the jobs instead of launching the generation itself.
static bool multi_can_fuse_generation(const vh_context *c, int32_t num_bins, int32_t capacity)
{
    uint32_t parts = (uint32_t)grid_for((size_t)capacity, 256 * 4);
    if (parts < 1) parts = 1;
    // (not the walk-free launch, flatten_variant 4: it has no 17 us walk for the generating workgroups' chain to end inside, and it
    // leaves most of the chip to a generation launched beside it -- one rank, frames/s: separate launches 96.0 k, fused with 3 / 4 / 6 / 2
    // groups per workgroup 90.2 / 87.6 / 82.3 / 72.6 k, 1 group 39.7 k; profiles/r05_fused_generation_ab.txt, box 9)
    return c->pipelineShards && c->fp.bucketSize <= kMaxPipelinedBucket && !c->viewBlocks && !(c->fp.flags & kFlagOverflow) &&
           c->packetFormat == VH_PACKET_U16 && !(c->fp.allocBand > 0.0f) && c->flattenVariant != kWalkIndexed &&
           serial_launch_pays(c, (uint32_t)num_bins * parts + (uint32_t)grid_for(c->numEntries, kFlattenThreads * kEntriesPerLaneShort));
}

// ... and does it pay?  A generating workgroup's chain (6 tiles, ~11 us beside a walk) has to end inside the launch: it does where the
// shard's walk is long enough to hide it.  One rank, one camera, 640x480, frames/s fused against separate launches
// (tools/r05_small_shard.py, profiles/r05_fused_by_table_size.txt): 2^21 buckets of 5 entries 30.5 / 29.2 k, 2^20 50.2 / 47.0 k, 2^19
// 67.0 / 66.7 k, 2^18 74.9 / 89.2 k, 2^17 77.1 / 93.0 k (4 tiles per workgroup: 84.7 / 90.4 k and 85.4 / 93.7 k) -- so above 2^19 buckets,
// i.e. a walk of more than 60 MB (~10 us).  C2 cut 2, 4 or 8 ways generates with launches of its own; C5's 2^21 buckets per rank fuse.
static bool multi_fusing_pays(const vh_context *c)
{
    return (size_t)c->numEntries * sizeof(VoxelEntry) > ((size_t)60 << 20);
}

static int vh_apply_frames_batch_gen(vh_context *c, int32_t batch, const int32_t *d_bins, int32_t num_bins,
                                     int32_t capacity, int32_t bin_stride, int32_t frame_stride, int32_t num_cams,
                                     const float *d_packets, size_t packet_stride, size_t packet_frame_stride, const GenJob *jobs);

extern "C" int vh_apply_frames_batch(vh_context *c, int32_t batch, const int32_t *d_bins, int32_t num_bins,
                                     int32_t capacity, int32_t bin_stride, int32_t frame_stride, int32_t num_cams,
                                     const float *d_packets, size_t packet_stride, size_t packet_frame_stride)
{
    return vh_apply_frames_batch_gen(c, batch, d_bins, num_bins, capacity, bin_stride, frame_stride, num_cams, d_packets, packet_stride,
                                     packet_frame_stride, nullptr);
}

// jobs (nullable): one fused generation job per frame of the batch (GenJob, vh_shard.hip), riding in the launch of that frame
static int vh_apply_frames_batch_gen(vh_context *c, int32_t batch, const int32_t *d_bins, int32_t num_bins,
                                     int32_t capacity, int32_t bin_stride, int32_t frame_stride, int32_t num_cams,
                                     const float *d_packets, size_t packet_stride, size_t packet_frame_stride, const GenJob *jobs)
{
    if (!c || !d_bins || !d_packets || batch <= 0 || num_bins <= 0 || capacity < 2 || num_cams <= 0 ||
        num_cams > VH_MAX_CAMERAS)
        return fail(VH_ERR_INVALID_ARGUMENT, "bad argument");
    const size_t dense = packet_units(c);
    const bool perBatch = frame_stride == VH_BIN_PER_BATCH;            // one bin per source for all frames of the batch
    if (perBatch && batch > (int32_t)VH_MAX_CAMERAS) return fail(VH_ERR_INVALID_ARGUMENT, "a per-batch bin holds at most 32 frames");
    if (frame_stride == 0) frame_stride = capacity;
    if (bin_stride == 0) bin_stride = perBatch ? capacity : batch * frame_stride;
    if (packet_frame_stride == 0) packet_frame_stride = dense;
    if (packet_stride == 0) packet_stride = (size_t)batch * packet_frame_stride;
    if ((!perBatch && (frame_stride < capacity || bin_stride < batch * frame_stride)) || (perBatch && bin_stride < capacity) || packet_frame_stride < dense ||
        packet_stride < (size_t)batch * packet_frame_stride)
        return fail(VH_ERR_INVALID_ARGUMENT, "bad stride");
    DeviceGuard guard(c->device);
    { const int frc = flush_single_pending(c); if (frc != VH_OK) return frc; }    // (a pending multi-camera half rides along)
    {
        const int rc = ensure_candidates(c, (size_t)num_bins * (size_t)(capacity - 1));
        if (rc != VH_OK) return rc;
    }
    uint32_t parts = (uint32_t)grid_for((size_t)capacity, 256 * 4);
    if (parts < 1) parts = 1;
    const uint32_t scanBlocks = (uint32_t)grid_for(c->numEntries, kFlattenThreads * kEntriesPerLane);
    const uint32_t commitBlocks = (uint32_t)c->commitBlocks;
    // One launch per multi-camera frame (frame_multi_pipelined_kernel, vh_shard.hip): the commit + TSDF update of frame
    // b ride in the launch of frame b+1.  pipeline_shards 1: a last launch serves the batch's last frame, B + 1 launches
    // instead of 2 B; 2: that half stays pending across calls (launch_multi_pipelined) and rides in the first launch of
    // the next batch -- B launches -- or in the flush any observer does first.
    // Same conditions as the single-camera pipeline (bucketSize <= 16, not a view table; with the overflow list the frames are
    // serialised inside the launch).
    if (c->pipelineShards && c->fp.bucketSize <= kMaxPipelinedBucket && !c->viewBlocks &&
        serial_launch_pays(c, (uint32_t)num_bins * parts + (uint32_t)grid_for(c->numEntries, kFlattenThreads * kEntriesPerLaneShort))) {
        int rc = ensure_pipeline_buffers(c);
        if (rc != VH_OK) return rc;
        if (!c->maskBuf2) VH_HIP(hipMalloc((void **)&c->maskBuf2, sizeof(uint32_t) * c->numEntries));
        MultiPending &mp = c->multiPend;
        // a pending half of another shape (camera count, packet layout) cannot share a launch with this batch's frames
        if (mp.active && (mp.numCams != num_cams || mp.packetStride != packet_stride || mp.packetFormat != c->packetFormat) &&
            (rc = flush_multi_pending(c)) != VH_OK)
            return rc;
        MultiBatch mb;
        mb.bins = reinterpret_cast<const int4 *>(d_bins); mb.packets = d_packets;
        mb.numBins = num_bins; mb.capacity = capacity; mb.binStride = bin_stride; mb.frameStride = perBatch ? 0 : frame_stride;
        mb.perBatch = perBatch;
        mb.numCams = num_cams; mb.packetStride = packet_stride; mb.packetFrameStride = packet_frame_stride;
        int b = 0;
        while (b < batch) {
            bool doNew = true;
            // at the epoch wrap vh_reset_mutexes clears the claim words, which the pending frame still needs: it is
            // served by a launch of its own first
            if (mp.active && c->fp.epoch >= kMaxClaimEpoch) doNew = false;
            if ((rc = launch_multi_pipelined(c, doNew ? &mb : nullptr, b, doNew && jobs ? jobs + b : nullptr)) != VH_OK) return rc;
            if (doNew) {
                if (b == 0 && c->multiFirstEvent) VH_HIP(hipEventRecord(reinterpret_cast<hipEvent_t>(c->multiFirstEvent), c->stream));
                ++b;
            }
        }
        if (c->pipelineShards < 2 && (rc = flush_multi_pending(c)) != VH_OK) return rc;
        VH_HIP(hipGetLastError());
        return VH_OK;
    }
    { const int frc = flush_multi_pending(c); if (frc != VH_OK) return frc; }
    for (int b = 0; b < batch; ++b) {
        int rc = vh_reset_mutexes(c);
        if (rc != VH_OK) return rc;
        const int4 *bins = reinterpret_cast<const int4 *>(d_bins) + (perBatch ? (size_t)0 : (size_t)frame_stride * b);
        const float *packets = d_packets + packet_frame_stride * b;
        rc = launch(c, kPhaseFrameScanClaim, frame_multi_scan_claim_kernel,
                    dim3((uint32_t)num_bins * parts + scanBlocks), dim3(256), c->fp, c->dp, bins, capacity, bin_stride,
                    (uint32_t)num_bins, parts, (uint32_t)c->numEntries, num_cams, packets, packet_stride,
                    c->fusedParity, claim_span(c, (uint32_t)num_bins * parts, scanBlocks),
                    claim_ratio((uint32_t)num_bins * parts, claim_span(c, (uint32_t)num_bins * parts, scanBlocks)), perBatch ? b : -1);
        if (rc == VH_OK)
            rc = c->packetFormat == VH_PACKET_U16
                     ? launch(c, kPhaseFrameCommitIntegrate, frame_multi_commit_integrate_kernel<true>,
                              dim3(commitBlocks + (uint32_t)c->integrateGrid), dim3(256), c->fp, c->dp, num_cams,
                              packets, packet_stride, commitBlocks, c->fusedParity)
                     : launch(c, kPhaseFrameCommitIntegrate, frame_multi_commit_integrate_kernel<false>,
                              dim3(commitBlocks + (uint32_t)c->integrateGrid), dim3(256), c->fp, c->dp, num_cams,
                              packets, packet_stride, commitBlocks, c->fusedParity);
        if (rc != VH_OK) return rc;
        c->fusedParity ^= 1;
        c->compactArmed = false;
        c->occupiedCounter = kCompactCount;
        c->foldA = -1;
        if (c->profiling) c->profiledFrames += 1;
    }
    if (c->multiFirstEvent) VH_HIP(hipEventRecord(reinterpret_cast<hipEvent_t>(c->multiFirstEvent), c->stream));
    VH_HIP(hipGetLastError());
    return VH_OK;
}

extern "C" int vh_insert_bins(vh_context *c, const int32_t *d_bins, int32_t num_bins, int32_t capacity,
                              int32_t bin_stride)
{
    if (bin_stride == 0) bin_stride = capacity;
    if (!c || !d_bins || num_bins <= 0 || capacity < 2 || bin_stride < capacity)
        return fail(VH_ERR_INVALID_ARGUMENT, "bad argument");
    if (c->epochTotal == 0) return fail(VH_ERR_INVALID_ARGUMENT, "vh_reset_mutexes must start the frame");
    DeviceGuard guard(c->device);
    { const int frc = flush_pending(c); if (frc != VH_OK) return frc; }
    {
        const int rc = ensure_candidates(c, (size_t)num_bins * (size_t)(capacity - 1));
        if (rc != VH_OK) return rc;
    }
    int gx = grid_for((size_t)capacity, 256 * 4);
    if (gx < 1) gx = 1;
    int rc = launch(c, kPhaseClaim, claim_bins_kernel, dim3(gx, num_bins), dim3(256), c->fp, c->dp,
                    reinterpret_cast<const int4 *>(d_bins), capacity, bin_stride);
    if (rc == VH_OK) rc = launch(c, kPhaseCommit, alloc_commit_kernel, dim3(32), dim3(256), c->fp, c->dp);
    if (rc != VH_OK) return rc;
    c->compactArmed = true;
    VH_HIP(hipGetLastError());
    return VH_OK;
}

extern "C" int vh_integrate_packets(vh_context *c, int32_t num_cams, const float *d_packets, size_t packet_stride)
{
    const size_t dense = c ? packet_units(c) : 0;
    if (packet_stride == 0) packet_stride = dense;
    if (!c || !d_packets || num_cams <= 0 || num_cams > VH_MAX_CAMERAS || packet_stride < dense)
        return fail(VH_ERR_INVALID_ARGUMENT, "bad argument");
    DeviceGuard guard(c->device);
    { const int frc = flush_pending(c); if (frc != VH_OK) return frc; }
    const size_t stride = packet_stride;
    if (!c->compactArmed)
        VH_HIP(hipMemsetAsync(c->dp.counters + kCompactCount, 0, sizeof(int32_t), c->stream));
    c->compactArmed = false;
    c->occupiedCounter = kCompactCount;
    c->foldA = -1;
    int rc = launch(c, kPhaseFlatten, flatten_multi_kernel,
                    dim3(grid_for(c->numEntries, kFlattenThreads * kEntriesPerLane)), dim3(kFlattenThreads), c->fp,
                    c->dp, (uint32_t)c->numEntries, num_cams, d_packets, stride);
    if (rc == VH_OK)
        rc = c->packetFormat == VH_PACKET_U16
                 ? launch(c, kPhaseIntegrate, integrate_multi_kernel<true>, dim3(c->integrateGrid), dim3(256), c->fp, c->dp,
                          num_cams, d_packets, stride)
                 : launch(c, kPhaseIntegrate, integrate_multi_kernel<false>, dim3(c->integrateGrid), dim3(256), c->fp,
                          c->dp, num_cams, d_packets, stride);
    if (rc != VH_OK) return rc;
    if (c->profiling) c->profiledFrames += 1;
    VH_HIP(hipGetLastError());
    return VH_OK;
}
