// vh_view.hip -- raycast over bucket-range shards: every shard hands the blocks a view can
// touch to the rank that renders the view, which raycasts a private "view table" holding
// exactly those blocks (SURVEY.md 8(e) "replicate the compact table + visible blocks").
// Part of libvoxelhash_hip.so (gfx950); included by vh_kernels.hip after vh_device.h.
#pragma once

namespace vh {

constexpr int kMaxViewsPerLaunch = 16;
constexpr int kViewRecordBytes = 4112;                 // {int32 pos[3], 0, 512 x {sdf, weight}}
constexpr int kViewRecordVoxels = kViewRecordBytes / 8;

// Prepared on the host (vh_api.hip make_view_frustum; oracle: vho_view_frustum):
// f[0..11] rows 0..2 of world->camera, f[12..15] plane slopes (left, right, top, bottom),
// f[16..19] their thresholds -(r*sqrt(1+a^2)), f[20] zLo, f[21] zHi.
struct ViewFrustum { float f[22]; };
struct ViewSet { ViewFrustum v[kMaxViewsPerLaunch]; };

// Conservative test "a ray of the view can sample a voxel of block pos": block centre within
// 7 voxels (> the 6.93-voxel half diagonal) of every bounding plane of the view pyramid.
// Same operations in the same order as the oracle's vho_view_holds_block.
__device__ __forceinline__ bool view_holds_block(const FrameParams &fp, const float *__restrict__ f, const int *pos)
{
    float c[3], pc[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) c[a] = ((float)(int)((uint32_t)pos[a] * 8u) + 3.5f) * fp.voxelSize;
#pragma unroll
    for (int r = 0; r < 3; ++r) pc[r] = f[4 * r + 0] * c[0] + f[4 * r + 1] * c[1] + f[4 * r + 2] * c[2] + f[4 * r + 3];
    if (!(pc[2] >= f[20] && pc[2] <= f[21])) return false;
    if (!(pc[0] - f[12] * pc[2] >= f[16])) return false;
    if (!(f[13] * pc[2] - pc[0] >= f[17])) return false;
    if (!(pc[1] - f[14] * pc[2] >= f[18])) return false;
    if (!(f[15] * pc[2] - pc[1] >= f[19])) return false;
    return true;
}

// The same preparation on the device (vh_export_views_fixed: the view poses arrive by all-gather and
// never visit the host): rows 0..2 of the cofactor inverse (the 12 of its 16 sums the test needs, same
// order as the host's invert4x4 / cuda_SimpleMatrixUtil.h:944-1069), slopes, thresholds.  One lane per view.
struct ViewCamera { float fx, fy, cx, cy, tMin, tMax; int32_t width, height; };

__device__ __forceinline__ void invert4x4_device(const float *e, float *out)
{
    // adjugate by cofactors, each the sum of six signed triple products in the reference's order
    const signed char cof[16][6][4] = {
        {{+1,5,10,15},{-1,5,11,14},{-1,9,6,15},{+1,9,7,14},{+1,13,6,11},{-1,13,7,10}},
        {{-1,1,10,15},{+1,1,11,14},{+1,9,2,15},{-1,9,3,14},{-1,13,2,11},{+1,13,3,10}},
        {{+1,1,6,15},{-1,1,7,14},{-1,5,2,15},{+1,5,3,14},{+1,13,2,7},{-1,13,3,6}},
        {{-1,1,6,11},{+1,1,7,10},{+1,5,2,11},{-1,5,3,10},{-1,9,2,7},{+1,9,3,6}},
        {{-1,4,10,15},{+1,4,11,14},{+1,8,6,15},{-1,8,7,14},{-1,12,6,11},{+1,12,7,10}},
        {{+1,0,10,15},{-1,0,11,14},{-1,8,2,15},{+1,8,3,14},{+1,12,2,11},{-1,12,3,10}},
        {{-1,0,6,15},{+1,0,7,14},{+1,4,2,15},{-1,4,3,14},{-1,12,2,7},{+1,12,3,6}},
        {{+1,0,6,11},{-1,0,7,10},{-1,4,2,11},{+1,4,3,10},{+1,8,2,7},{-1,8,3,6}},
        {{+1,4,9,15},{-1,4,11,13},{-1,8,5,15},{+1,8,7,13},{+1,12,5,11},{-1,12,7,9}},
        {{-1,0,9,15},{+1,0,11,13},{+1,8,1,15},{-1,8,3,13},{-1,12,1,11},{+1,12,3,9}},
        {{+1,0,5,15},{-1,0,7,13},{-1,4,1,15},{+1,4,3,13},{+1,12,1,7},{-1,12,3,5}},
        {{-1,0,5,11},{+1,0,7,9},{+1,4,1,11},{-1,4,3,9},{-1,8,1,7},{+1,8,3,5}},
        {{-1,4,9,14},{+1,4,10,13},{+1,8,5,14},{-1,8,6,13},{-1,12,5,10},{+1,12,6,9}},
        {{+1,0,9,14},{-1,0,10,13},{-1,8,1,14},{+1,8,2,13},{+1,12,1,10},{-1,12,2,9}},
        {{-1,0,5,14},{+1,0,6,13},{+1,4,1,14},{-1,4,2,13},{-1,12,1,6},{+1,12,2,5}},
        {{+1,0,5,10},{-1,0,6,9},{-1,4,1,10},{+1,4,2,9},{+1,8,1,6},{-1,8,2,5}},
    };
    float inv[16];
    for (int o = 0; o < 16; ++o) {
        float acc = 0.0f;
        for (int k = 0; k < 6; ++k) {
            float t = (e[cof[o][k][1]] * e[cof[o][k][2]]) * e[cof[o][k][3]];
            if (cof[o][k][0] < 0) t = -t;
            acc = (k == 0) ? t : acc + t;
        }
        inv[o] = acc;
    }
    const float det = e[0] * inv[0] + e[1] * inv[4] + e[2] * inv[8] + e[3] * inv[12];
    const float detr = 1.0f / det;
    for (int i = 0; i < 16; ++i) out[i] = inv[i] * detr;
}

__global__ void view_frustum_kernel(const float *__restrict__ poses, int32_t numViews, ViewCamera cam, float voxelSize,
                                    ViewSet *__restrict__ out)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= numViews) return;
    float inv[16];
    invert4x4_device(poses + 16 * (size_t)v, inv);
    float *f = out->v[v].f;
    for (int i = 0; i < 12; ++i) f[i] = inv[i];
    const float r = 7.0f * voxelSize;
    const float a[4] = {(0.0f - cam.cx) / cam.fx, ((float)(cam.width - 1) - cam.cx) / cam.fx,
                        (0.0f - cam.cy) / cam.fy, ((float)(cam.height - 1) - cam.cy) / cam.fy};
    for (int i = 0; i < 4; ++i) {
        f[12 + i] = a[i];
        f[16 + i] = -(r * __builtin_sqrtf(1.0f + a[i] * a[i]));
    }
    f[20] = cam.tMin - r;
    f[21] = cam.tMax + r;
}

// One walk over the shard's entries for up to 16 views (the same strided ptr-dword stream as
// the flatten walk): a live entry is tested against every view and its index appended to the
// list of each view that holds it (one atomic per wave and view that has hits).
__device__ __forceinline__ void view_select(const FrameParams &fp, const DevPtrs &dp, uint32_t numEntries,
                                            const ViewSet &vs, int32_t numViews, int32_t *__restrict__ lists,
                                            int32_t capacity, int32_t *__restrict__ counts)
{
    const uint32_t tile = blockIdx.x * (kFlattenThreads * kEntriesPerLane);
    int32_t ptrs[kEntriesPerLane];
    walk_load_tile(fp, dp, numEntries, blockIdx.x, ptrs);
    bool any = false;
#pragma unroll
    for (int j = 0; j < kEntriesPerLane; ++j) any |= ptrs[j] != VH_FREE_BLOCK;
    if (__ballot(any) == 0ull) return;
    const int lane = threadIdx.x & (kWave - 1);
#pragma unroll
    for (int j = 0; j < kEntriesPerLane; ++j) {
        const uint32_t e = tile + j * kFlattenThreads + threadIdx.x;
        uint32_t seen = 0;
        if (ptrs[j] != VH_FREE_BLOCK) {
            const VoxelEntry ent = dp.table[e];
            for (int v = 0; v < numViews; ++v)
                if (view_holds_block(fp, vs.v[v].f, ent.pos)) seen |= 1u << v;
        }
        if (__ballot(seen != 0u) == 0ull) continue;
        for (int v = 0; v < numViews; ++v) {
            const bool hit = (seen >> v) & 1u;
            const unsigned long long mask = __ballot(hit);
            if (mask == 0ull) continue;
            int base = 0;
            const int leaderLane = __ffsll((long long)mask) - 1;
            if (lane == leaderLane) base = atomicAdd(counts + v, __popcll(mask));
            base = __shfl(base, leaderLane);
            const int slot = base + __popcll(mask & ((1ull << lane) - 1ull));
            if (hit && slot < capacity) lists[(size_t)v * capacity + slot] = (int32_t)e;
        }
    }
}

__global__ __launch_bounds__(kFlattenThreads) void view_select_kernel(const FrameParams fp, const DevPtrs dp,
                                                                      uint32_t numEntries, const ViewSet vs,
                                                                      int32_t numViews, int32_t *__restrict__ lists,
                                                                      int32_t capacity, int32_t *__restrict__ counts)
{
    view_select(fp, dp, numEntries, vs, numViews, lists, capacity, counts);
}

// the same with the prepared views in device memory (view_frustum_kernel)
__global__ __launch_bounds__(kFlattenThreads) void view_select_mem_kernel(const FrameParams fp, const DevPtrs dp,
                                                                          uint32_t numEntries,
                                                                          const ViewSet *__restrict__ vs,
                                                                          int32_t numViews, int32_t *__restrict__ lists,
                                                                          int32_t capacity, int32_t *__restrict__ counts)
{
    __shared__ ViewSet s;
    for (int i = threadIdx.x; i < (int)(sizeof(ViewSet) / 4); i += blockDim.x)
        reinterpret_cast<float *>(&s)[i] = reinterpret_cast<const float *>(vs)[i];
    __syncthreads();
    view_select(fp, dp, numEntries, s, numViews, lists, capacity, counts);
}

// Records of view v follow those of views 0..v-1 without gaps (the caller sends them with
// per-destination counts); blockIdx.y = view, workgroups stride over its selected entries and
// copy key + 4 KiB of voxels, 16 bytes per lane.
// fixedSlots: view v's records start at v * capacity instead (fixed-size exchange, no counts on the host).
__global__ __launch_bounds__(256) void view_pack_kernel(const DevPtrs dp, const int32_t *__restrict__ lists,
                                                        const int32_t *__restrict__ counts, int32_t capacity,
                                                        uint8_t *__restrict__ records, int32_t fixedSlots)
{
    const int v = blockIdx.y;
    size_t first = 0;
    if (fixedSlots) first = (size_t)v * (size_t)capacity;
    else for (int u = 0; u < v; ++u) first += (size_t)min(counts[u], capacity);
    const int n = min(counts[v], capacity);
    // fixed slots: the first record's spare header word carries the view's count, so the counts travel
    // inside the payload (an empty view gets a key-less header there)
    if (fixedSlots && n == 0 && blockIdx.x == 0 && threadIdx.x == 0)
        *reinterpret_cast<int4 *>(records + first * kViewRecordBytes) =
            make_int4(VH_POS_SENTINEL, VH_POS_SENTINEL, VH_POS_SENTINEL, counts[v]);
    for (int b = blockIdx.x; b < n; b += gridDim.x) {
        const VoxelEntry e = dp.table[lists[(size_t)v * capacity + b]];
        uint8_t *rec = records + (first + (size_t)b) * kViewRecordBytes;
        if (threadIdx.x == 0)
            *reinterpret_cast<int4 *>(rec) = make_int4(e.pos[0], e.pos[1], e.pos[2], (fixedSlots && b == 0) ? counts[v] : 0);
        reinterpret_cast<float4 *>(rec + 16)[threadIdx.x] =
            reinterpret_cast<const float4 *>(dp.blocks + (size_t)e.ptr)[threadIdx.x];
    }
}

// View table: the buckets the previous import touched (list in compactMask) go back to empty.
__global__ __launch_bounds__(256) void view_clear_kernel(const FrameParams fp, const DevPtrs dp, int32_t prevCount)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= prevCount) return;
    const uint32_t h = dp.compactMask[i];
    if (h == ~0u) return;                    // an unused slot of a fixed-capacity import
    VoxelEntry free_;
    free_.pos[0] = free_.pos[1] = free_.pos[2] = VH_POS_SENTINEL;
    free_.ptr = VH_FREE_BLOCK;
    free_.offset = 0;
    for (uint32_t s = 0; s < fp.bucketSize; ++s) dp.table[(size_t)h * fp.bucketSize + s] = free_;
    dp.claim[h] = 0ull;
}

// Record i becomes an entry of its bucket (slots are handed out by a per-bucket fill count kept
// in the otherwise unused claim word, so the entries of a bucket form a prefix, which
// lookup_block relies on); ptr addresses the voxels inside the record buffer itself.
// counts != nullptr (vh_import_views): `count` = numSources * capacity slots, source s's records sit in
// slots [s * capacity, s * capacity + min(counts[s], capacity)); the other slots are skipped (their
// compactMask entry says "no bucket") and what a source selected beyond the capacity is counted as lost.
__global__ __launch_bounds__(256) void view_import_kernel(const FrameParams fp, const DevPtrs dp,
                                                          const uint8_t *__restrict__ records, int32_t count,
                                                          const int32_t *__restrict__ counts, int32_t capacity)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    if (capacity > 0) {
        const int src = i / capacity, k = i - src * capacity;
        // the source's count: from the device array, or from the spare header word of its first record
        const int have = counts ? counts[src]
                                : reinterpret_cast<const int4 *>(records + (size_t)src * capacity * kViewRecordBytes)->w;
        if (k == 0 && have > capacity) atomicAdd(dp.counters + kBinOverflow, have - capacity);
        if (k >= min(have, capacity)) { dp.compactMask[i] = ~0u; return; }
    }
    const int4 k = *reinterpret_cast<const int4 *>(records + (size_t)i * kViewRecordBytes);
    const uint32_t h = hash_block(k.x, k.y, k.z, fp.numBuckets);
    dp.compactMask[i] = h;
    const uint32_t slot = atomicAdd(reinterpret_cast<unsigned int *>(dp.claim + h), 1u);
    if (slot >= fp.bucketSize) {
        // more records than slots in this bucket: with the overflow list on (the shards' table has
        // chains) the record is placed by view_import_overflow_kernel, otherwise it is lost and counted
        if (fp.flags & kFlagOverflow) {
            const uint32_t q = (uint32_t)atomicAdd(dp.counters + kCandCount, 1);
            if (q < dp.candCapacity) { dp.candTarget[q] = (uint32_t)i; return; }
        }
        atomicAdd(dp.counters + kBinOverflow, 1);
        return;
    }
    VoxelEntry e;
    e.pos[0] = k.x; e.pos[1] = k.y; e.pos[2] = k.z;
    e.ptr = i * kViewRecordVoxels + 2;
    e.offset = 0;
    dp.table[(size_t)h * fp.bucketSize + slot] = e;
    atomicOr(dp.bucketBits + (h >> 5), 1u << (h & 31u));
    const uint32_t hm = macro_hash(k.x >> 2, k.y >> 2, k.z >> 2);
    atomicOr(dp.macroBits + (hm >> 5), 1u << (hm & 31u));
}

// The records that found their bucket full, one after the other (they are few): each goes to a free
// slot among the kLookAhead-1 slots behind its home bucket (not a bucket's last slot) and to the front
// of the home bucket's chain, like an insertion into the shards' table (vh_alloc.hip), without the
// locks -- nothing else runs on a view table.  Which slot a record gets does not matter: the view
// table only has to answer lookups.
__global__ void view_import_overflow_kernel(const FrameParams fp, const DevPtrs dp, const uint8_t *__restrict__ records)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const int n = min(dp.counters[kCandCount], (int)dp.candCapacity);
    const uint32_t total = owned_entries(fp), bs = fp.bucketSize;
    for (int q = 0; q < n; ++q) {
        const uint32_t i = dp.candTarget[q];
        const int4 k = *reinterpret_cast<const int4 *>(records + (size_t)i * kViewRecordBytes);
        const uint32_t h = hash_block(k.x, k.y, k.z, fp.numBuckets);
        const uint32_t last = h * bs + bs - 1u;
        uint32_t links = 0, at = last;
        bool ended = false;
        for (uint32_t iter = 0; iter < fp.listSize; ++iter) {
            const VoxelEntry curr = dp.table[at];
            if (curr.offset == 0) { ended = true; break; }
            at = chain_slot(last, curr.offset, total);
            ++links;
        }
        uint32_t target = ~0u;
        if (ended && fp.listSize >= 2u && links + 1u <= fp.listSize - 1u)
            for (int j = 1; j < kLookAhead; ++j) {
                const uint32_t s = chain_slot(last, j, total);
                if (s % bs == bs - 1u) continue;
                if (dp.table[s].ptr == VH_FREE_BLOCK) { target = s; break; }
            }
        if (target == ~0u) {
            atomicAdd(dp.counters + kBinOverflow, 1);
            continue;
        }
        VoxelEntry e;
        e.pos[0] = k.x; e.pos[1] = k.y; e.pos[2] = k.z;
        e.ptr = (int)i * kViewRecordVoxels + 2;
        e.offset = dp.table[last].offset;
        dp.table[target] = e;
        dp.table[last].offset = (int)(target >= last ? target - last : target + total - last);
        const uint32_t tb = target / bs;
        dp.compactMask[i] = tb;                      // the next import clears this bucket too (the home bucket is listed by its own records)
        atomicOr(dp.bucketBits + (tb >> 5), 1u << (tb & 31u));
        const uint32_t hm = macro_hash(k.x >> 2, k.y >> 2, k.z >> 2);
        atomicOr(dp.macroBits + (hm >> 5), 1u << (hm & 31u));
    }
    dp.counters[kCandCount] = 0;
}

}  // namespace vh
