// vh_api_model.hip -- C-ABI, the model: deletion / garbage collection, queries, dump and snapshot, options and profiling.
// Included by vh_api.hip (same translation unit: shares fail(), VH_HIP, DeviceGuard, launch()).

// ---------------------------------------------------------------------------
// block deletion / garbage collection
// ---------------------------------------------------------------------------
static int sweep_and_release(vh_context *c)
{
    int rc;
    if (c->fp.flags & kFlagOverflow) {
        rc = launch(c, kPhaseGc, gc_sweep_overflow_a_kernel, dim3(256), dim3(256), c->fp, c->dp);
        if (rc == VH_OK) rc = launch(c, kPhaseGc, gc_sweep_overflow_b_kernel, dim3(256), dim3(256), c->fp, c->dp);
    } else {
        rc = launch(c, kPhaseGc, gc_sweep_kernel, dim3(256), dim3(256), c->fp, c->dp);
    }
    if (rc != VH_OK) return rc;
    rc = launch(c, kPhaseGc, gc_release_kernel, dim3(1024), dim3(256), c->dp);
    if (rc != VH_OK) return rc;
    rc = launch(c, kPhaseGc, gc_finish_kernel, dim3(1), dim3(1), c->dp, c->occupiedCounter);
    if (rc != VH_OK) return rc;
    if (c->profiling) c->times.gc_calls += 1;
    c->params.numOccupiedBlocks = 0;
    VH_HIP(hipGetLastError());
    return VH_OK;
}

extern "C" int vh_delete_blocks(vh_context *c, const int32_t *d_keys, int32_t n)
{
    if (!c || (!d_keys && n > 0) || n < 0) return fail(VH_ERR_INVALID_ARGUMENT, "bad argument");
    if (c->viewBlocks) return fail(VH_ERR_INVALID_ARGUMENT, "a view table owns no blocks");
    DeviceGuard guard(c->device);
    { const int frc = settle(c); if (frc != VH_OK) return frc; }
    {   // the sweep list is built under a fresh lock epoch (with the wrap handling of the frame's epochs)
        const int rc = vh_reset_mutexes(c);
        if (rc != VH_OK) return rc;
    }
    if (n > 0) {
        const int rc = launch(c, kPhaseGc, gc_mark_keys_kernel, dim3((unsigned)grid_for((size_t)n, 256)), dim3(256), c->fp,
                              c->dp, reinterpret_cast<const int4 *>(d_keys), n);
        if (rc != VH_OK) return rc;
    }
    return sweep_and_release(c);
}

extern "C" int vh_garbage_collect(vh_context *c, float sdf_threshold)
{
    VH_TRACE("vh_garbage_collect");
    if (!c) return fail(VH_ERR_INVALID_ARGUMENT, "null context");
    if (c->viewBlocks) return fail(VH_ERR_INVALID_ARGUMENT, "a view table owns no blocks");
    DeviceGuard guard(c->device);
    { const int frc = settle(c); if (frc != VH_OK) return frc; }
    {
        const int rc = vh_reset_mutexes(c);
        if (rc != VH_OK) return rc;
    }
    const int rc = launch(c, kPhaseGc, gc_identify_kernel, dim3(2048), dim3(256), c->fp, c->dp, c->occupiedCounter,
                          sdf_threshold);
    if (rc != VH_OK) return rc;
    return sweep_and_release(c);
}

// ---------------------------------------------------------------------------
// queries
// ---------------------------------------------------------------------------
// A workgroup of a serialised launch that gives up waiting (wait_commit_done, vh_frame.hip) drops its claim / walk work of that
// frame: the model is then wrong, and every call has returned VH_OK.  So wherever the library synchronises with the host
// anyway, and a serialised launch has been queued since the last look, the counter is read (4 bytes) and a new timeout is an
// ERROR there -- once; the context falls back to two launches per overflow-list frame at the same moment.
static int check_spin_timeouts(vh_context *c)
{
    if (!c->serialQueued) return VH_OK;
    int32_t n = 0;
    VH_HIP(hipMemcpyAsync(&n, c->dp.counters + kSpinTimeouts, sizeof n, hipMemcpyDeviceToHost, c->stream));
    VH_HIP(hipStreamSynchronize(c->stream));
    c->serialQueued = false;
    if ((uint32_t)n == c->spinSeen) return VH_OK;
    c->spinSeen = (uint32_t)n;
    c->serialFallback = true;
    return fail(VH_ERR_TIMEOUT, "workgroups of a serialised one-launch frame gave up waiting for the pending frame's commit phase "
                                "(vh_counters.spin_timeouts): frames queued since the last synchronisation have lost work");
}

extern "C" int vh_synchronize(vh_context *c)
{
    if (!c) return fail(VH_ERR_INVALID_ARGUMENT, "null context");
    DeviceGuard guard(c->device);
    { const int frc = settle(c); if (frc != VH_OK) return frc; }
    VH_HIP(hipStreamSynchronize(c->stream));
    return check_spin_timeouts(c);
}

extern "C" int vh_get_counters(vh_context *c, vh_counters *out)
{
    if (!c || !out) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    DeviceGuard guard(c->device);
    { const int frc = flush_pending(c); if (frc != VH_OK) return frc; }
    int32_t h[kNumCounters];
    VH_HIP(hipMemcpyAsync(h, c->dp.counters, sizeof h, hipMemcpyDeviceToHost, c->stream));
    VH_HIP(hipStreamSynchronize(c->stream));
    out->occupied = h[c->occupiedCounter];
    out->heap_counter = h[kHeapCounter];
    out->allocated_total = (uint32_t)h[kAllocatedTotal];
    out->heap_exhausted = (uint32_t)h[kHeapExhausted];
    out->candidates = (uint32_t)h[kLastCandidates];
    out->epoch = c->epochTotal;
    out->bin_overflow = (uint32_t)h[kBinOverflow];
    out->freed_total = (uint32_t)h[kFreedTotal];
    out->last_freed = (uint32_t)h[kLastFreed];
    out->cand_overflow = (uint32_t)h[kCandOverflow];
    out->spin_timeouts = (uint32_t)h[kSpinTimeouts];
    if (out->spin_timeouts) c->serialFallback = true;       // (overflow-list frames: two launches each from now on)
    c->spinSeen = out->spin_timeouts;                       // (reported here: the next synchronisation does not fail for it again)
    c->serialQueued = false;
    c->params.numOccupiedBlocks = (uint32_t)h[c->occupiedCounter];
    return VH_OK;
}

extern "C" int vh_get_params(vh_context *c, HashTableParams *out)
{
    if (!c || !out) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    *out = c->params;
    return VH_OK;
}

extern "C" int vh_get_device_pointers(vh_context *c, PtrContainer *out)
{
    if (!c || !out) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    { DeviceGuard guard(c->device); const int frc = settle(c); if (frc != VH_OK) return frc; }   // the dense compact list
    out->d_heap = c->dp.heap;
    out->d_hashTable = c->dp.table;
    out->d_compactifiedHashTable = c->compactHome;          // (= c->dp.compact after settle(): stable for the context's life)
    // stable too: the claim array of creation.  Pipelined frames stake the claims of consecutive lock epochs
    // alternately in this array and in a second one (every word carries its epoch in its top 10 bits).
    out->d_hashTableBucketMutex = reinterpret_cast<uint64_t *>(c->claimBuf[0] ? c->claimBuf[0] : c->dp.claim);
    out->d_SDFBlocks = c->dp.blocks;
    out->d_heapCounter = c->dp.counters + kHeapCounter;
    out->d_compactifiedHashCounter = c->dp.counters + kCompactCount;
    return VH_OK;
}

static int download_range(vh_context *c, int which, size_t offset, void *dst, size_t bytes)
{
    if (!c || !dst) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    { DeviceGuard fguard(c->device); const int frc = flush_pending(c); if (frc != VH_OK) return frc; }
    const char *src = nullptr;
    size_t avail = 0;
    switch (which) {
        case VH_BUF_HASH_TABLE: src = (const char *)c->dp.table; avail = sizeof(VoxelEntry) * c->numEntries; break;
        case VH_BUF_COMPACT: src = (const char *)c->dp.compact; avail = sizeof(VoxelEntry) * c->numEntries; break;
        case VH_BUF_SDF_BLOCKS:
            src = (const char *)(c->viewBlocks ? c->viewBlocks : c->dp.blocks);
            avail = c->viewBlocks ? (size_t)c->viewCount * sizeof(vh_view_record)
                                  : sizeof(Voxel) * (size_t)c->params.numVoxelBlocks * kBlockVoxels;
            break;
        case VH_BUF_HEAP: src = (const char *)c->dp.heap; avail = sizeof(uint32_t) * (size_t)c->params.numVoxelBlocks; break;
        default: return fail(VH_ERR_INVALID_ARGUMENT, "unknown buffer id");
    }
    if (offset > avail || bytes > avail - offset) return fail(VH_ERR_INVALID_ARGUMENT, "download past the end of the buffer");
    DeviceGuard guard(c->device);
    { const int frc = settle(c); if (frc != VH_OK) return frc; }
    VH_HIP(hipMemcpyAsync(dst, src + offset, bytes, hipMemcpyDeviceToHost, c->stream));
    VH_HIP(hipStreamSynchronize(c->stream));
    return check_spin_timeouts(c);         // (what was downloaded is the model of frames that may have lost work: say so)
}

extern "C" int vh_download(vh_context *c, int which, void *dst, size_t bytes)
{
    return download_range(c, which, 0, dst, bytes);
}

extern "C" int vh_download_range(vh_context *c, int which, size_t offset_bytes, void *dst, size_t bytes)
{
    return download_range(c, which, offset_bytes, dst, bytes);
}

// ---------------------------------------------------------------------------
// model dump / checkpoint
// ---------------------------------------------------------------------------
// SDFRenderer::printSDFdata (SDFRenderer.cpp:71-110), the reference's only on-disk artefact:
// the occupied count, then per compact entry "pos / ptr / offset" and 512 sdf values with 4
// decimals.  Faithful to a quirk of the original: the 512 values printed for entry i are voxels
// [512*i, 512*i+512) of the volume (it reads the first count*512 voxels, :85-87), NOT the block
// the entry's ptr names.
extern "C" int vh_dump_sdf_text(vh_context *c, const char *path)
{
    if (!c || !path) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    vh_counters k;
    int rc = vh_get_counters(c, &k);
    if (rc != VH_OK) return rc;
    const size_t n = (size_t)(k.occupied > 0 ? k.occupied : 0);
    std::vector<VoxelEntry> entries(n);
    const size_t nvox = std::min(n * kBlockVoxels, (size_t)c->params.numVoxelBlocks * kBlockVoxels);
    std::vector<Voxel> vox(n * kBlockVoxels, Voxel{0.0f, 0.0f});
    if (n) {
        if ((rc = vh_download(c, VH_BUF_COMPACT, entries.data(), n * sizeof(VoxelEntry))) != VH_OK) return rc;
        if ((rc = vh_download(c, VH_BUF_SDF_BLOCKS, vox.data(), nvox * sizeof(Voxel))) != VH_OK) return rc;
    }
    FILE *f = std::fopen(path, "w");
    if (!f) return fail(VH_ERR_INVALID_ARGUMENT, "cannot open the dump file");
    std::fprintf(f, "numOccupiedBlocks from GL :%zu\n", n);                                  // :95
    std::fprintf(f, "\nSDFs \n\n");                                                           // :100
    for (size_t i = 0; i < n; ++i) {
        const VoxelEntry &e = entries[i];
        std::fprintf(f, "%zu) : pos : (%d, %d, %d) ptr = %d offset = %d\n", i, e.pos[0], e.pos[1], e.pos[2], e.ptr,
                     e.offset);                                                               // :102-103
        for (int j = 0; j < kBlockVoxels; ++j) std::fprintf(f, "%.4f\t", vox[i * kBlockVoxels + j].sdf);   // :104-106
        std::fprintf(f, "\n\n\n");
    }
    std::fclose(f);
    return VH_OK;
}

// Binary snapshot: header, hash table, heap, then the 4 KiB block of every allocated entry in
// table order.  Enough to continue fusing after vh_load_snapshot as if never interrupted.
struct SnapshotHeader {
    char magic[8];                 // "VHSNAP01"
    HashTableParams params;
    int32_t width, height, semantics;
    uint32_t bucketLo, bucketHi;
    int32_t heapCounter;
    uint32_t allocatedTotal, heapExhausted, epoch;
    uint64_t numEntries, numAllocated;
    float proj[9];
};

extern "C" int vh_save_snapshot(vh_context *c, const char *path)
{
    if (!c || !path) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    vh_counters k;
    int rc = vh_get_counters(c, &k);
    if (rc != VH_OK) return rc;
    std::vector<VoxelEntry> table(c->numEntries);
    std::vector<uint32_t> heap(c->params.numVoxelBlocks);
    if ((rc = vh_download(c, VH_BUF_HASH_TABLE, table.data(), table.size() * sizeof(VoxelEntry))) != VH_OK) return rc;
    if ((rc = vh_download(c, VH_BUF_HEAP, heap.data(), heap.size() * sizeof(uint32_t))) != VH_OK) return rc;
    SnapshotHeader h{};
    std::memcpy(h.magic, "VHSNAP01", 8);
    h.params = c->params;
    h.width = c->fp.width; h.height = c->fp.height; h.semantics = c->fp.semantics;
    h.bucketLo = c->fp.bucketLo; h.bucketHi = c->fp.bucketHi;
    h.heapCounter = k.heap_counter; h.allocatedTotal = k.allocated_total; h.heapExhausted = k.heap_exhausted;
    h.epoch = c->epochTotal;
    h.numEntries = c->numEntries;
    std::memcpy(h.proj, c->fp.proj, sizeof h.proj);
    for (const VoxelEntry &e : table) h.numAllocated += e.ptr != VH_FREE_BLOCK;
    // written next to the target and renamed over it at the end: a crash mid-save never leaves a
    // truncated file under the final name
    const std::string tmp = std::string(path) + ".partial";
    FILE *f = std::fopen(tmp.c_str(), "wb");
    if (!f) return fail(VH_ERR_INVALID_ARGUMENT, "cannot open the snapshot file");
    bool ok = std::fwrite(&h, sizeof h, 1, f) == 1;
    ok = ok && std::fwrite(table.data(), sizeof(VoxelEntry), table.size(), f) == table.size();
    ok = ok && std::fwrite(heap.data(), sizeof(uint32_t), heap.size(), f) == heap.size();
    DeviceGuard guard(c->device);
    std::vector<Voxel> block(kBlockVoxels);
    for (const VoxelEntry &e : table) {
        if (e.ptr == VH_FREE_BLOCK || !ok) continue;
        if (hipMemcpy(block.data(), c->dp.blocks + e.ptr, sizeof(Voxel) * kBlockVoxels, hipMemcpyDeviceToHost) !=
            hipSuccess) { ok = false; break; }
        ok = std::fwrite(block.data(), sizeof(Voxel), kBlockVoxels, f) == (size_t)kBlockVoxels;
    }
    ok = (std::fflush(f) == 0) && ok;
    ok = (std::fclose(f) == 0) && ok;
    if (ok) ok = std::rename(tmp.c_str(), path) == 0;
    if (!ok) {
        (void)std::remove(tmp.c_str());
        return fail(VH_ERR_HIP, "snapshot write failed");
    }
    return VH_OK;
}

// Empties the model (table, heap, counters, bitmaps, volume) on the context's stream.
static hipError_t reset_model(vh_context *c)
{
    hipStream_t s = c->stream;
    const int g = 2048;
    reset_table_kernel<<<g, 256, 0, s>>>(c->dp.table, c->numEntries);
    reset_heap_kernel<<<g, 256, 0, s>>>(c->dp.heap, c->params.numVoxelBlocks);
    hipError_t e = hipMemsetAsync(c->dp.claim, 0, sizeof(unsigned long long) * (size_t)c->ownedBuckets, s);
    if (e == hipSuccess) e = hipMemsetAsync(c->dp.bucketBits, 0, sizeof(uint32_t) * (((size_t)c->ownedBuckets + 31) / 32), s);
    if (e == hipSuccess) e = hipMemsetAsync(c->dp.macroBits, 0, kMacroBits / 8, s);
    if (e == hipSuccess) e = hipMemsetAsync(c->dp.blocks, 0, sizeof(Voxel) * (size_t)c->params.numVoxelBlocks * kBlockVoxels, s);
    int32_t counters[kNumCounters] = {0};
    counters[kHeapCounter] = (int32_t)c->params.numVoxelBlocks - 1;
    if (e == hipSuccess) e = hipMemcpyAsync(c->dp.counters, counters, sizeof counters, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    c->fp.epoch = 0;
    c->fusedParity = 0;
    c->pipePending = false;
    c->compactArmed = false;
    c->occupiedCounter = kCompactCount;
    c->foldA = -1;
    if (e == hipSuccess && c->claimBuf[1])
        e = hipMemset(c->claimBuf[1 - (c->dp.claim == c->claimBuf[1] ? 1 : 0)], 0, sizeof(unsigned long long) * (size_t)c->ownedBuckets);
    return e;
}

// The file is read and validated on the host -- header against this context, every entry, the
// heap, the pool partition, the file size -- BEFORE the first byte of device state changes: a
// truncated, corrupt or mismatched file leaves the live model untouched.  Only the voxel payload is
// streamed afterwards; should reading it fail then (an I/O error after the size check), the model
// is reset to empty rather than left half-loaded.
extern "C" int vh_load_snapshot(vh_context *c, const char *path)
{
    if (!c || !path) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    if (c->viewBlocks) return fail(VH_ERR_INVALID_ARGUMENT, "a view table owns no blocks");
    FILE *f = std::fopen(path, "rb");
    if (!f) return fail(VH_ERR_INVALID_ARGUMENT, "cannot open the snapshot file");
    struct Closer { FILE *f; ~Closer() { std::fclose(f); } } closer{f};
    SnapshotHeader h;
    if (std::fread(&h, sizeof h, 1, f) != 1 || std::memcmp(h.magic, "VHSNAP01", 8) != 0)
        return fail(VH_ERR_INVALID_ARGUMENT, "not a snapshot file");
    const HashTableParams &p = c->params;
    const bool match = h.numEntries == c->numEntries && h.params.numVoxelBlocks == p.numVoxelBlocks &&
                       h.params.numBuckets == p.numBuckets && h.params.bucketSize == p.bucketSize &&
                       h.params.voxelBlockSize == p.voxelBlockSize && h.bucketLo == c->fp.bucketLo &&
                       h.bucketHi == c->fp.bucketHi && h.width == c->fp.width && h.height == c->fp.height &&
                       h.semantics == c->fp.semantics &&
                       std::memcmp(&h.params.voxelSize, &p.voxelSize, sizeof(float)) == 0 &&
                       std::memcmp(&h.params.truncation, &p.truncation, sizeof(float)) == 0 &&
                       std::memcmp(&h.params.integrationWeightMax, &p.integrationWeightMax, sizeof(float)) == 0;
    if (!match) return fail(VH_ERR_INVALID_ARGUMENT, "snapshot does not match this context");
    const int64_t pool = (int64_t)p.numVoxelBlocks;
    if (h.heapCounter < -1 || (int64_t)h.heapCounter >= pool)
        return fail(VH_ERR_INVALID_ARGUMENT, "snapshot is corrupt: heap counter out of range");
    std::vector<VoxelEntry> table(c->numEntries);
    std::vector<uint32_t> heap(p.numVoxelBlocks);
    if (std::fread(table.data(), sizeof(VoxelEntry), table.size(), f) != table.size() ||
        std::fread(heap.data(), sizeof(uint32_t), heap.size(), f) != heap.size())
        return fail(VH_ERR_INVALID_ARGUMENT, "snapshot is truncated");
    // every block id is either referenced by exactly one entry or on the free part of the heap
    std::vector<uint8_t> seen(p.numVoxelBlocks, 0);
    uint64_t allocated = 0;
    for (const VoxelEntry &e : table) {
        if (e.ptr == VH_FREE_BLOCK) continue;
        if (e.ptr < 0 || e.ptr % kBlockVoxels != 0 || (int64_t)(e.ptr / kBlockVoxels) >= pool || seen[e.ptr / kBlockVoxels])
            return fail(VH_ERR_INVALID_ARGUMENT, "snapshot is corrupt: bad or duplicate block pointer");
        seen[e.ptr / kBlockVoxels] = 1;
        ++allocated;
        if (e.offset != 0 && !(c->fp.flags & kFlagOverflow))
            return fail(VH_ERR_INVALID_ARGUMENT, "snapshot holds overflow chains: set the option overflow_list before loading it");
        if (e.offset < 0 || e.offset >= kLookAhead)
            return fail(VH_ERR_INVALID_ARGUMENT, "snapshot is corrupt: chain offset out of range");
    }
    for (int64_t i = 0; i <= (int64_t)h.heapCounter; ++i) {
        if ((int64_t)heap[i] >= pool || seen[heap[i]])
            return fail(VH_ERR_INVALID_ARGUMENT, "snapshot is corrupt: free list overlaps the allocated blocks");
        seen[heap[i]] = 1;
    }
    if (allocated != h.numAllocated || (int64_t)allocated + (int64_t)h.heapCounter + 1 != pool)
        return fail(VH_ERR_INVALID_ARGUMENT, "snapshot is corrupt: blocks and free list do not partition the pool");
    const long payload_at = std::ftell(f);
    if (payload_at < 0 || std::fseek(f, 0, SEEK_END) != 0) return fail(VH_ERR_INVALID_ARGUMENT, "snapshot is unreadable");
    const long file_end = std::ftell(f);
    if (file_end < 0 || (uint64_t)(file_end - payload_at) != allocated * sizeof(Voxel) * kBlockVoxels)
        return fail(VH_ERR_INVALID_ARGUMENT, "snapshot is truncated or has trailing bytes");
    if (std::fseek(f, payload_at, SEEK_SET) != 0) return fail(VH_ERR_INVALID_ARGUMENT, "snapshot is unreadable");

    // ---- from here on the device state changes ----
    DeviceGuard guard(c->device);
    { const int frc = flush_pending(c); if (frc != VH_OK) return frc; }
    hipError_t e = reset_model(c);
    const size_t words = ((size_t)c->ownedBuckets + 31) / 32;
    std::vector<uint32_t> bits(words, 0u), macro(kMacroBits / 32, 0u);
    std::vector<Voxel> block(kBlockVoxels);
    bool ok = true;
    for (size_t i = 0; ok && e == hipSuccess && i < table.size(); ++i) {
        if (table[i].ptr == VH_FREE_BLOCK) continue;
        const size_t bucket = i / p.bucketSize;
        bits[bucket >> 5] |= 1u << (bucket & 31);
        const uint32_t hm = ((((uint32_t)(table[i].pos[0] >> 2)) * 73856093u) ^ (((uint32_t)(table[i].pos[1] >> 2)) * 19349669u) ^
                             (((uint32_t)(table[i].pos[2] >> 2)) * 83492791u)) & (kMacroBits - 1u);
        macro[hm >> 5] |= 1u << (hm & 31);
        ok = std::fread(block.data(), sizeof(Voxel), kBlockVoxels, f) == (size_t)kBlockVoxels;
        if (ok) e = hipMemcpy(c->dp.blocks + table[i].ptr, block.data(), sizeof(Voxel) * kBlockVoxels, hipMemcpyHostToDevice);
    }
    int32_t counters[kNumCounters] = {0};
    counters[kHeapCounter] = h.heapCounter;
    counters[kAllocatedTotal] = (int32_t)h.allocatedTotal;
    counters[kHeapExhausted] = (int32_t)h.heapExhausted;
    if (ok && e == hipSuccess) e = hipMemcpy(c->dp.table, table.data(), sizeof(VoxelEntry) * table.size(), hipMemcpyHostToDevice);
    if (ok && e == hipSuccess) e = hipMemcpy(c->dp.heap, heap.data(), sizeof(uint32_t) * heap.size(), hipMemcpyHostToDevice);
    if (ok && e == hipSuccess) e = hipMemcpy(c->dp.bucketBits, bits.data(), sizeof(uint32_t) * words, hipMemcpyHostToDevice);
    if (ok && e == hipSuccess) e = hipMemcpy(c->dp.macroBits, macro.data(), kMacroBits / 8, hipMemcpyHostToDevice);
    if (ok && e == hipSuccess) e = hipMemcpy(c->dp.counters, counters, sizeof counters, hipMemcpyHostToDevice);
    if (!ok || e != hipSuccess) {
        (void)reset_model(c);                       // never leave a half-loaded model behind
        return !ok ? fail(VH_ERR_INVALID_ARGUMENT, "snapshot payload could not be read; the model was reset to empty")
                   : fail(VH_ERR_HIP, "snapshot upload failed; the model was reset to empty", e);
    }
    // pose, projection: the snapshot's; geometry and fusion constants were checked equal above
    std::memcpy(c->params.global_transform, h.params.global_transform, sizeof c->fp.T);
    std::memcpy(c->params.inv_global_transform, h.params.inv_global_transform, sizeof c->fp.Tinv);
    std::memcpy(c->fp.T, h.params.global_transform, sizeof c->fp.T);
    std::memcpy(c->fp.Tinv, h.params.inv_global_transform, sizeof c->fp.Tinv);
    std::memcpy(c->fp.proj, h.proj, sizeof h.proj);
    c->params.numOccupiedBlocks = 0;
    c->epochTotal = h.epoch;         // (frames have been integrated into this model: build-time options stay locked)
    return VH_OK;
}

extern "C" int vh_set_option(vh_context *c, const char *name, int value)
{
    if (c) { DeviceGuard fguard(c->device); const int frc = flush_pending(c); if (frc != VH_OK) return frc; }
    if (!c || !name) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    if (std::strcmp(name, "flatten_variant") == 0) {
        if (value != kWalkStridedBallot && value != kWalkIndexed)
            return fail(VH_ERR_INVALID_ARGUMENT, "flatten_variant: 3 (the reference's walk) or 4 (occupancy index)");
        c->flattenVariant = value;
        return VH_OK;
    }
    if (std::strcmp(name, "walk_nt") == 0) {
        c->fp.flags = value ? (c->fp.flags | kFlagWalkNt) : (c->fp.flags & ~kFlagWalkNt);
        return VH_OK;
    }
#ifdef VH_DEBUG_SKIP_ROLES      // diagnostics builds only (make EXTRA=-DVH_DEBUG_SKIP_ROLES): the check costs the product launch a scalar load per workgroup
    if (std::strcmp(name, "debug_skip_roles") == 0 && value >= 0 && value < 32) {
        c->debugSkipRoles = value;
        // (a shard's multi-camera launch: bit 16 = the claim role finds its bins empty -- what the probes of received keys cost it)
        if (value & 16) c->fp.flags |= kFlagDebugNoProbe; else c->fp.flags &= ~kFlagDebugNoProbe;
        return VH_OK;
    }
#endif
    if (std::strcmp(name, "spin_limit") == 0 && value >= 0) { c->spinLimit = (uint32_t)value; return VH_OK; }
    if (std::strcmp(name, "pipeline_overflow") == 0 && value >= 0 && value <= 2) {
        DeviceGuard g(c->device);
        const int frc = flush_pending(c);               // (the form of the next frame may change)
        if (frc != VH_OK) return frc;
        c->pipelineOverflow = value;
        return VH_OK;
    }
    if (std::strcmp(name, "gen_frames_per_launch") == 0 && value >= 1 && value <= kGenBatch) { c->genFramesPerLaunch = value; return VH_OK; }
    if (std::strcmp(name, "pipeline_shards") == 0) {
        if (value < 0 || value > 2) return fail(VH_ERR_INVALID_ARGUMENT, "pipeline_shards: 0, 1 or 2");
        if (value < 2) { DeviceGuard g(c->device); const int frc = flush_multi_pending(c); if (frc != VH_OK) return frc; }
        c->pipelineShards = value;
        return VH_OK;
    }
    if (std::strcmp(name, "pipeline") == 0) {
        c->pipeline = value != 0;
        return VH_OK;                // (a pending frame was flushed at the top of this call)
    }
    if (std::strcmp(name, "overflow_list") == 0) {
        // a table is built with the list or without it: the two keep different invariants (holes vs prefix)
        if (c->epochTotal != 0 && ((c->fp.flags & kFlagOverflow) != 0u) != (value != 0))
            return fail(VH_ERR_INVALID_ARGUMENT, "overflow_list must be chosen before the first frame");
        if (c->fp.listSize < 2 && value) return fail(VH_ERR_INVALID_ARGUMENT, "attachedLinkedListSize must be at least 2");
        c->fp.flags = value ? (c->fp.flags | kFlagOverflow) : (c->fp.flags & ~kFlagOverflow);
        return VH_OK;
    }
    if (std::strcmp(name, "band_mode") == 0 && (value == VH_BAND_RAY || value == VH_BAND_NORMAL_DDA || value == VH_BAND_RAY_DDA)) {
        c->fp.flags &= ~(kFlagBandDda | kFlagBandRayDda);
        if (value == VH_BAND_NORMAL_DDA) c->fp.flags |= kFlagBandDda;
        if (value == VH_BAND_RAY_DDA) c->fp.flags |= kFlagBandRayDda;
        return VH_OK;
    }
    if (std::strcmp(name, "depth_truncation") == 0) {
        c->fp.flags = value ? (c->fp.flags | kFlagDepthTruncation) : (c->fp.flags & ~kFlagDepthTruncation);
        return VH_OK;
    }
    if (std::strcmp(name, "weight_sample") == 0) {
        c->fp.flags = value ? (c->fp.flags | kFlagWeightSample) : (c->fp.flags & ~kFlagWeightSample);
        return VH_OK;
    }
    if (std::strcmp(name, "cand_capacity") == 0 && value > 0) {     // test hook: a smaller candidate list
        c->dp.candCapacity = std::min<uint32_t>((uint32_t)value, c->candAllocated);
        return VH_OK;
    }
    if (std::strcmp(name, "integrate_grid") == 0 && value > 0) { c->integrateGrid = value; return VH_OK; }
    if (std::strcmp(name, "fused_frame") == 0) { c->fusedFrame = value; return VH_OK; }
    if (std::strcmp(name, "raycast_beam") == 0 && value >= 0 && value <= 3) { c->raycastBeam = value; return VH_OK; }
    if (std::strcmp(name, "raycast_mode") == 0 && (value == VH_RAYCAST_DDA || value == VH_RAYCAST_FIXED_STEP)) {
        c->raycastMode = value;
        return VH_OK;
    }
    if (std::strcmp(name, "packet_format") == 0 && (value == VH_PACKET_F32 || value == VH_PACKET_U16)) {
        c->packetFormat = value;
        return VH_OK;
    }
    if (std::strcmp(name, "commit_blocks") == 0 && value > 0) { c->commitBlocks = value; return VH_OK; }
    return fail(VH_ERR_INVALID_ARGUMENT, "unknown option");
}

extern "C" int vh_set_profiling(vh_context *c, int enabled)
{
    if (!c) return fail(VH_ERR_INVALID_ARGUMENT, "null context");
    c->profiling = enabled != 0;
    return VH_OK;
}

// Folds the pending event pairs into c->times (synchronises the stream) and returns them to the pool.
static int accumulate_times(vh_context *c)
{
    VH_HIP(hipStreamSynchronize(c->stream));
    // every pair leaves `timed` before it is looked at and goes back to the pool whatever hipEventElapsedTime
    // says, so a failure part-way never leaves a pair listed twice (launch() would reuse it while still listed,
    // drop_events() destroy it twice)
    std::vector<TimedLaunch> done;
    done.swap(c->timed);
    hipError_t firstError = hipSuccess;
    for (auto &t : done) {
        float ms = 0;
        const hipError_t e = hipEventElapsedTime(&ms, t.start, t.stop);
        c->eventPool.emplace_back(t.start, t.stop);
        if (e != hipSuccess) { if (firstError == hipSuccess) firstError = e; continue; }
        switch (t.phase) {
            case kPhaseClaim: c->times.alloc_claim_ms += ms; break;
            case kPhaseCommit: c->times.alloc_commit_ms += ms; break;
            case kPhaseFlatten: c->times.flatten_ms += ms; break;
            case kPhaseIntegrate: c->times.integrate_ms += ms; break;
            case kPhaseRaycast: c->times.raycast_ms += ms; c->times.raycast_launches += 1; break;     // (the split form: one pair around its three launches)
            case kPhaseFrameScanClaim: c->times.frame_scan_claim_ms += ms; break;
            case kPhaseFrameCommitIntegrate: c->times.frame_commit_integrate_ms += ms; break;
            case kPhaseViewExport: c->times.view_export_ms += ms; break;
            case kPhaseViewImport: c->times.view_import_ms += ms; break;
            case kPhaseGc: c->times.gc_ms += ms; break;
            case kPhaseRaycastBounds: c->times.render_blocks_ms += ms; break;
            case kPhaseFramePipelined: c->times.frame_pipelined_ms += ms; break;
            default: break;
        }
    }
    c->times.launches += c->profiledFrames;
    c->profiledFrames = 0;
    if (firstError != hipSuccess) return fail(VH_ERR_HIP, "hipEventElapsedTime", firstError);
    return VH_OK;
}

extern "C" int vh_get_kernel_times(vh_context *c, vh_kernel_times *out, int reset)
{
    if (!c || !out) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    DeviceGuard guard(c->device);
    const int rc = accumulate_times(c);
    if (rc != VH_OK) return rc;
    *out = c->times;
    if (reset) c->times = vh_kernel_times{};
    return VH_OK;
}

#ifdef VH_CLAIM_STAMPS
extern "C" int vh_debug_set_claim_stamps(void *d_stamps)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(vh::g_claimStamps), &d_stamps, sizeof d_stamps) == hipSuccess ? VH_OK : VH_ERR_HIP;
}
#endif

// test hook (tests/test_gpu_concurrency.py): `workgroups` workgroups of 256 lanes that do nothing but stay resident for
// `microseconds` -- a chip that is busy with somebody else's long kernel -- on `stream` (any stream of the context's device)
__global__ __launch_bounds__(256) void debug_occupy_kernel(unsigned long long ticks)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
extern "C" int vh_debug_occupy(vh_context *c, void *stream, int32_t workgroups, int32_t microseconds)
{
    if (!c || workgroups < 1 || microseconds < 0) return fail(VH_ERR_INVALID_ARGUMENT, "bad argument");
    DeviceGuard guard(c->device);
    debug_occupy_kernel<<<dim3((unsigned)workgroups), dim3(256), 0, (hipStream_t)stream>>>((unsigned long long)microseconds * 100ull);
    VH_HIP(hipGetLastError());
    return VH_OK;
}

// diagnostics hook (tools/raycast_stamps.py): the DDA raycast writes {start, end (s_memrealtime, 100 MHz), steps, patch}
// per wave into d_stamps (4 uint64 per wave, waves in workgroup order); NULL switches it off
extern "C" int vh_debug_set_raycast_stamps(vh_context *c, void *d_stamps)
{
    if (!c) return fail(VH_ERR_INVALID_ARGUMENT, "null context");
    c->raycastStamps = d_stamps;
    return VH_OK;
}

// test hook: scalar helpers evaluated on the device (8 int32 per point:
// block x,y,z, hash, inFrustum, screen x,y, f2i_rz(w))
extern "C" int vh_debug_eval(vh_context *c, const vh_float4 *d_points, int32_t n, int32_t *d_out)
{
    if (!c || !d_points || !d_out || n < 0) return fail(VH_ERR_INVALID_ARGUMENT, "bad argument");
    DeviceGuard guard(c->device);
    if (n == 0) return VH_OK;
    debug_eval_kernel<<<grid_for((size_t)n, 256), 256, 0, c->stream>>>(c->fp, reinterpret_cast<const float4 *>(d_points), n,
                                                                       d_out);
    VH_HIP(hipGetLastError());
    return VH_OK;
}
