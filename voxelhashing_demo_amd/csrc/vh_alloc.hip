// vh_alloc.hip -- allocBlocks: per-pixel block keys, wave-level run dedup, bucket probe + epoch-stamped claim (phase 1),
// commit of the winners (phase 2), key generation for the multi-GPU exchange.
// Part of libvoxelhash_hip.so (gfx950); included by vh_kernels.hip after vh_device.h.
#pragma once

namespace vh {

// ---------------------------------------------------------------------------
// bucket probe shared by the claim kernels
// ---------------------------------------------------------------------------
// Reads the bucket of `key` the way insertVoxelEntry scans it (VoxelUtils.cu:436-456):
// present -> nothing to do; otherwise, if a free slot exists, stake a claim.
// Allocated entries always form a prefix of the bucket (insertions take the
// first free slot, deletion closes the gap: vh_gc.hip), so "present anywhere" equals the
// reference's in-order scan.
// The same with the overflow list on (kFlagOverflow; oracle: insert_entry_overflow).  A bucket's entries
// no longer form a prefix (deletion leaves holes), so all slots are scanned; the chain behind the
// bucket's last slot is walked with the reference's lookup loop; a key whose home bucket is full
// looks for a free slot among the kLookAhead-1 slots behind it (never another bucket's last slot,
// which heads that bucket's own chain) and then needs BOTH buckets: it stakes its claim on both and
// commits only if it holds both (the reference locks the parent bucket, then the new one,
// VoxelUtils.cu:472-482; both stay locked for the frame).
// (Round 4 built the wave's version of this probe -- eight lanes per key: the bucket's slots read side by side and judged by
// a ballot, the chain followed in step, the nine look-ahead slots read together, the group's first lane staking the claim;
// slot- and link-exact on the whole overflow suite -- and measured it against this one, same box: the two-launch frame of C2
// with the list on 16.5 -> 17.0 us, with a 10 cm band 19.9 -> 22.5 us (profiles/r04_ab_overflow_coop.txt; branch
// wip/overflow-probe-wave).  The keys of a wave are probed side by side already, one per lane; serving them eight at a time
// shortens a key's chain of reads but makes eight rounds of it.  Not adopted.)
__device__ __forceinline__ void probe_and_claim_overflow(const FrameParams &fp, const DevPtrs &dp, int kx, int ky, int kz,
                                                         uint32_t h, uint32_t rank, int candCounter)
{
    const uint32_t local = h - fp.bucketLo, bs = fp.bucketSize, n = owned_entries(fp);
    const uint32_t start = local * bs, last = start + bs - 1u;
    bool has_free = false;
    for (uint32_t i = 0; i < bs; ++i) {
        const VoxelEntry e = dp.table[start + i];
        if (entry_is(e, kx, ky, kz)) return;
        has_free |= e.ptr == VH_FREE_BLOCK;
    }
    uint32_t links = 0, i = last;
    bool ended = false;
    for (uint32_t iter = 0; iter < fp.listSize; ++iter) {
        const VoxelEntry curr = dp.table[i];
        if (entry_is(curr, kx, ky, kz)) return;
        if (curr.offset == 0) { ended = true; break; }
        i = chain_slot(last, curr.offset, n);
        ++links;
    }
    uint32_t target = ~0u;
    if (!has_free) {
        if (!ended || fp.listSize < 2u || links + 1u > fp.listSize - 1u) return;    // chain at the reach of the lookup loop
        for (int j = 1; j < kLookAhead; ++j) {                                     // :475-478
            const uint32_t s = chain_slot(last, j, n);
            if (s % bs == bs - 1u) continue;
            if (dp.table[s].ptr == VH_FREE_BLOCK) { target = s; break; }
        }
        if (target == ~0u) return;
    }
    const uint32_t slot = (uint32_t)atomicAdd(dp.counters + candCounter, 1);
    if (slot >= dp.candCapacity) {
        atomicAdd(dp.counters + kCandOverflow, 1);
        return;
    }
    dp.candidates[slot] = make_int4(kx, ky, kz, (int)rank);
    dp.candTarget[slot] = target;
    atomicMax(dp.claim + local, claim_word(fp.epoch, rank, 0u, slot));
    if (target != ~0u) atomicMax(dp.claim + target / bs, claim_word(fp.epoch, rank, 0u, slot));
}

// Pipelined frames (vh_frame.hip): the frame whose commit phase runs CONCURRENTLY with this claim
// phase.  Its claim words and candidate list are final (they were written by the previous launch),
// so a bucket's pending insertion is known without looking at the slot that is being written: word
// of the previous epoch = {who: slot -> key, where: f}.  live = false: that frame's insertions are
// all refused (the heap cannot serve them all, see frame_pipelined_kernel) and the table is as it reads.
struct Pending {
    const unsigned long long *__restrict__ claim;     // nullptr: no frame in flight
    const int4 *__restrict__ cand;
    uint32_t epoch;
    bool live;
    int winnersCounter;                               // of the frame being claimed: counts its distinct buckets (-1: not counted)
    const uint32_t *__restrict__ filter;              // the pending frame's claim filter (below), or nullptr: every claim word is read
    uint32_t *filterNew;                              // the filter the frame being claimed fills, or nullptr
};
__device__ constexpr Pending kNoPending{nullptr, nullptr, 0u, false, -1, nullptr, nullptr};

// Claim filter (round 5).  "Is an insertion in flight into this bucket?" is asked by every probe of the claim phase and
// for every allocated entry the walk meets, and the answer sits in the pending frame's claim word: 8 bytes out of an array
// of 8 bytes per bucket (134 MB at 2^24 buckets) -- a round trip to HBM to learn, 99.9 % of the time, "nothing".  The
// buckets with an insertion in flight are the pending frame's winners: a few hundred at most.  So the claim phase also sets
// one bit per staked bucket in a filter of 2^17 bits (16 KB: L2-resident; indexed by the bucket's low bits, three of them
// rotating with the per-frame counter sets: filled by frame i, read by the launch of frame i+1, cleared by that of i+2);
// readers touch the claim word only behind a set bit.  A false positive reads the word as before: exactness is untouched.
constexpr uint32_t kPendFilterWords = 4096;
__device__ __forceinline__ bool pend_maybe(const Pending &pend, uint32_t local)
{
    return pend.filter == nullptr || ((pend.filter[(local >> 5) & (kPendFilterWords - 1u)] >> (local & 31u)) & 1u) != 0u;
}

// kEagerSlot: the bucket's first slot is requested together with the pending frame's claim word -- two independent addresses,
// one round trip instead of two; what it returns is not looked at when that slot is the one being written.  It pays where the
// probes answer from HBM or the claim tiles are long (same box, launch us without / with: C3 68.3 / 67.6, C5table 278.5 / 275.5,
// C2 with the band 22.75 / 22.2) and costs the cache-resident reference frame a little (C2 17.68 / 17.78): the callers choose.
// slot0 (nullable): the bucket's first slot as the caller has requested it already (claim_tile: before the frustum test, whose
// ~100 instructions then run under that round trip); implies what kEagerSlot does with it.
template <bool kEagerSlot = false>
__device__ __forceinline__ void probe_and_claim(const FrameParams &fp, const DevPtrs &dp, int kx, int ky, int kz,
                                                uint32_t h, uint32_t rank, int candCounter = kCandCount,
                                                const Pending &pend = kNoPending, const VoxelEntry *slot0 = nullptr)
{
    if (fp.flags & kFlagOverflow) {
        probe_and_claim_overflow(fp, dp, kx, ky, kz, h, rank, candCounter);
        return;
    }
    const uint32_t local = h - fp.bucketLo;
    const VoxelEntry *bucket = dp.table + (size_t)local * fp.bucketSize;
    VoxelEntry e0{};
    if (slot0) e0 = *slot0;
    else if (kEagerSlot) e0 = bucket[0];
    // the insertion in flight into this bucket, if any: it takes the bucket's first free slot (pf)
    uint32_t pf = ~0u;
    int4 pk = make_int4(0, 0, 0, 0);
    if (pend.claim && pend.live && pend_maybe(pend, local)) {
        const unsigned long long w = pend.claim[local];
        if (claim_epoch(w) == pend.epoch) {
            pf = claim_f(w);
            pk = pend.cand[claim_slot(w)];
        }
    }
    uint32_t firstFree = ~0u;
    for (uint32_t i = 0; i < fp.bucketSize; ++i) {
        if (i == pf) {                   // being written right now: never read, it WILL hold pk
            if (pk.x == kx && pk.y == ky && pk.z == kz) return;
            continue;
        }
        const VoxelEntry e = ((kEagerSlot || slot0) && i == 0u) ? e0 : bucket[i];
        if (e.ptr == VH_FREE_BLOCK) {
            firstFree = i;
            break;                       // prefix property: nothing allocated behind a free slot
        }
        if (e.pos[0] == kx && e.pos[1] == ky && e.pos[2] == kz) return;   // already allocated
    }
    if (firstFree == ~0u) return;        // bucket full: the key is dropped (no overflow list)
    // The record is reserved BEFORE the claim is staked: a contender that finds the candidate list
    // full must not hold a bucket's winning word (nobody would commit it).  It is counted instead.
    const uint32_t slot = (uint32_t)atomicAdd(dp.counters + candCounter, 1);
    if (slot >= dp.candCapacity) {
        atomicAdd(dp.counters + kCandOverflow, 1);
        return;
    }
    dp.candidates[slot] = make_int4(kx, ky, kz, (int)rank);
    const unsigned long long before = atomicMax(dp.claim + local, claim_word(fp.epoch, rank, firstFree, slot));
    if (pend.filterNew) atomicOr(pend.filterNew + ((local >> 5) & (kPendFilterWords - 1u)), 1u << (local & 31u));
    // first claim on this bucket in this epoch: one more entry the commit phase will insert
    if (pend.winnersCounter >= 0 && claim_epoch(before) != fp.epoch) atomicAdd(dp.counters + pend.winnersCounter, 1);
}

// ---------------------------------------------------------------------------
// allocBlocks, phase 1
// ---------------------------------------------------------------------------
// One 256-lane workgroup = one 16x16 block of the reference's launch grid (VoxelUtils.cu:610-611,
// 710-712), lane t = thread (t & 15, t >> 4) of it, so a lane's index inside its workgroup IS its
// position in the launch order that decides who wins a bucket: rank = tile*256 + t.
// A wave is a 16x4 pixel patch (four 256-byte row segments of the float4 vertex map per load
// instruction).  Neighbouring pixels almost always fall into the same 8^3 block, so each wave
// collapses equal keys before touching the table: a lane stays silent when the lane to its LEFT or
// the lane ABOVE it (both earlier in launch order) wants the same key.  By induction the earliest
// lane of every key in the wave survives, which is all the determinism rule needs; what survives
// redundantly only costs a probe.  ~300 k pixels become one or two thousand bucket probes (the
// row-only collapse of round 1 left ~9 k on C2 and ~80 k on C3).
// Truncation-band allocation (opt-in; commented out in the reference,
// VoxelUtils.cu:632-703): with fp.allocBand = b > 0 a pixel demands the blocks of
// 2*ceil(b/step)+1 points on its viewing ray at camera depths z + (k - half)*step, step = half
// a block edge; the middle sample is the surface point itself.  b = 0: that sample only.
struct PixelVertex {
    float4 v;
    float4 n;           // normal (kFlagBandDda only)
    int px, py;
    uint32_t rank;      // launch rank of the pixel = tile*256 + t
    bool valid;
};

__device__ __forceinline__ int band_samples(const FrameParams &fp, float &step)
{
    step = 4.0f * fp.voxelSize;
    if (!(fp.allocBand > 0.0f)) return 1;
    int half = (int)__builtin_ceilf(fp.allocBand / step);
    half = min(half, (kMaxBandSamples - 1) / 2);
    return 2 * half + 1;
}

// Where a pixel's vertex comes from: the float4 vertex map of the reference's interface, or the
// uint16 sensor image itself (vh_integrate_depth: calculateVertexPositions, CameraTrackingUtils.cu:
// 63-73, evaluated in place -- 2 bytes per pixel read instead of 16, no vertex map in memory).
struct VertexMap {
    const float4 *__restrict__ verts;
    const float4 *__restrict__ normals;      // preProcess's normal map (camera frame), read by the DDA band only; may be null
    __device__ __forceinline__ float4 normal(int idx) const
    {
        return normals ? normals[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __device__ __forceinline__ uint16_t raw(int) const { return 0; }
    // non-temporal: a vertex map is streamed once per frame, and keeping it out of the Infinity Cache
    // leaves more of the hash table there for the walk (launch 1: 17.5 -> 17.2 us)
    __device__ __forceinline__ float4 vertex(int idx, int, int) const
    {
        typedef float v4f __attribute__((ext_vector_type(4)));
        const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(verts + idx));   // one 16-byte load
        return make_float4(v.x, v.y, v.z, v.w);
    }
};

struct SensorImage {
    const uint16_t *__restrict__ depth;
    float k[9];              // K_inv, row-major
    float unit;              // 5000 = 1 m
    __device__ __forceinline__ float4 normal(int) const { return make_float4(0.f, 0.f, 0.f, 0.f); }   // no normal map
    __device__ __forceinline__ uint16_t raw(int idx) const { return depth[idx]; }
    __device__ __forceinline__ float4 vertex(int idx, int px, int py) const
    {
        const float d = (float)depth[idx] / unit;                                   // :64
        const float fx = (float)px, fy = (float)py;
        const float x = k[0] * fx + k[1] * fy + k[2] * 1.0f;                        // K_inv * (x, y, 1), :71
        const float y = k[3] * fx + k[4] * fy + k[5] * 1.0f;
        const float z = k[6] * fx + k[7] * fy + k[8] * 1.0f;
        return make_float4(x * d, y * d, z * d, 1.0f);                              // :73
    }
};

__device__ __forceinline__ uint32_t num_tiles(const FrameParams &fp)
{
    return (uint32_t)((fp.width + 15) >> 4) * (uint32_t)((fp.height + 15) >> 4);
}

// pixel of lane t (0..255) of launch tile `tile`
// outDepth / outRaw (optional): the pixel's camera z as a float plane (camera packets; the pipelined
// frame's private copy of what the TSDF update will gather) / the raw uint16 sensor value
template <class In>
__device__ __forceinline__ PixelVertex load_pixel(const FrameParams &fp, const In &in, uint32_t tile, uint32_t t,
                                                  float *__restrict__ outDepth, uint16_t *__restrict__ outRaw = nullptr)
{
    PixelVertex p{make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), 0, 0, (tile << 8) + t, false};
    const uint32_t tilesX = (uint32_t)(fp.width + 15) >> 4;
    const uint32_t by = tile / tilesX, bx = tile - by * tilesX;
    p.px = (int)(bx * 16u + (t & 15u));
    p.py = (int)(by * 16u + (t >> 4));
    if (p.px < fp.width && p.py < fp.height) {          // (a tile index past the grid gives py >= height)
        const int idx = p.py * fp.width + p.px;
        p.v = in.vertex(idx, p.px, p.py);
        if ((fp.flags & kFlagBandDda) && fp.allocBand > 0.0f) p.n = in.normal(idx);
        if (outDepth) outDepth[idx] = p.v.z;                             // camera-z plane of a camera packet
        if (outRaw) outRaw[idx] = in.raw(idx);
        p.valid = p.v.z != 0.0f;                                         // VoxelUtils.cu:621
    }
    return p;
}

// the same with the tile's position in the grid of tiles known (a caller that visits consecutive tiles divides once)
template <class In>
__device__ __forceinline__ PixelVertex load_pixel_at(const FrameParams &fp, const In &in, uint32_t tile, uint32_t bx, uint32_t by, uint32_t t,
                                                     float *__restrict__ outDepth, uint16_t *__restrict__ outRaw = nullptr)
{
    PixelVertex p{make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), 0, 0, (tile << 8) + t, false};
    p.px = (int)(bx * 16u + (t & 15u));
    p.py = (int)(by * 16u + (t >> 4));
    if (p.px < fp.width && p.py < fp.height) {
        const int idx = p.py * fp.width + p.px;
        p.v = in.vertex(idx, p.px, p.py);
        if ((fp.flags & kFlagBandDda) && fp.allocBand > 0.0f) p.n = in.normal(idx);
        if (outDepth) outDepth[idx] = p.v.z;
        if (outRaw) outRaw[idx] = in.raw(idx);
        p.valid = p.v.z != 0.0f;                                         // VoxelUtils.cu:621
    }
    return p;
}

struct SampleKey {
    int kx, ky, kz;
    bool leader;       // this lane must probe / emit the key
};

// The keys one pixel demands, in rank order (sample index k):
//   ray band (default)   2*half+1 points on the viewing ray, half-block steps, see above
//   DDA band             kFlagBandDda: every block the segment from p - b*n to p + b*n crosses, by a
//                        block DDA -- what the reference has commented out in allocBlocksKernel
//                        (VoxelUtils.cu:632-633 the two ends, :641-668 step / tMax / tDelta, :678-699 the
//                        walk).  p = the pixel's world point, n = its normal rotated into the world frame.
//                        The segment is parametrised over [0,1] (no normalisation, no square root); block
//                        k covers world [(8k - 0.5) * voxelSize, (8k + 7.5) * voxelSize) on an axis, which is
//                        what world2Block's rounding maps to k; ties as in :683-698; the walk ends at the end
//                        block, after 62 steps, or when the next crossing lies beyond the segment.  A pixel
//                        without a normal demands its surface block only.  Oracle: dda_keys.
struct BandWalk {
    int nS;             // ray band: number of samples (wave-uniform)
    float step;
    bool dda;           // wave-uniform
    // DDA state of this lane
    int cur[3], end[3], st[3];
    float tmax[3], tdelta[3];
    bool more;          // the walk has not ended
    // ray band: the two divisions of a sample by values that do not change from sample to sample (div_fixed, vh_device.h)
    float vsR1, zR1;
    bool vsOk, zOk;

    __device__ __forceinline__ void init(const FrameParams &fp, const PixelVertex &p)
    {
        nS = band_samples(fp, step);
        vsR1 = zR1 = 0.0f; vsOk = zOk = false;
        if (nS > 1) {            // (the reference's frame, one sample per pixel, keeps the plain divisions: nothing to hoist them out of)
            vsR1 = refined_rcp(fp.voxelSize); vsOk = fast_range(fp.voxelSize, 0x1p-40f, 0x1p40f);
            zR1 = refined_rcp(p.v.z); zOk = fast_range(p.v.z, 0x1p-40f, 0x1p40f);
        }
        const bool rayDda = (fp.flags & kFlagBandRayDda) && fp.allocBand > 0.0f;      // (wave-uniform)
        dda = ((fp.flags & kFlagBandDda) && fp.allocBand > 0.0f) || rayDda;
        more = false;
        if (!dda || !p.valid) return;
        float start[3], dir[3];
        // (round 5, measured and not kept: the fourteen divisions of this set-up -- two by z, six by voxelSize, two per axis by the
        // segment's component -- as div_fixed with shared refined reciprocals, the same bits: C2band's launch 22.3 -> 23.0 us, the
        // sample band 24.3 -> 24.3 -- each FixedDivisor::divide carries its plain-division fall-back; profiles/r05_band_div_fixed_ab.txt)
        if (rayDda) {
            // VH_BAND_RAY_DDA: the segment of the viewing ray between camera depths z - b and z + b (oracle: ray_dda_keys);
            // its two ends are the vertex scaled to those depths, through the pose as :622
            const float z = p.v.z, b = fp.allocBand;
            float s0 = z - b;
            if (!(s0 > 0.0f)) s0 = z;
            const float s1 = z + b;
            const float c0 = s0 / z, c1 = s1 / z;
            const float4 g0 = mat4_mul(fp.T, p.v.x * c0, p.v.y * c0, s0, p.v.w);
            const float4 g1 = mat4_mul(fp.T, p.v.x * c1, p.v.y * c1, s1, p.v.w);
            const float a0[3] = {g0.x, g0.y, g0.z}, a1[3] = {g1.x, g1.y, g1.z};
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                start[a] = a0[a];
                dir[a] = a1[a] - a0[a];
                cur[a] = voxel2block1(world2voxel1(a0[a], fp.voxelSize));
                end[a] = voxel2block1(world2voxel1(a1[a], fp.voxelSize));
            }
        } else {
        const float4 g = mat4_mul(fp.T, p.v.x, p.v.y, p.v.z, p.v.w);                 // :622, w as stored
        const int3_ sb = world2block(g.x, g.y, g.z, fp.voxelSize);
        cur[0] = sb.x; cur[1] = sb.y; cur[2] = sb.z;
        end[0] = sb.x; end[1] = sb.y; end[2] = sb.z;
        const float nx = p.n.x, ny = p.n.y, nz = p.n.z;
        if ((nx == 0.0f && ny == 0.0f && nz == 0.0f) || nx != nx || ny != ny || nz != nz) return;   // surface block only
        const float gw[3] = {g.x, g.y, g.z};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float nw = fp.T[4 * a + 0] * nx + fp.T[4 * a + 1] * ny + fp.T[4 * a + 2] * nz;
            start[a] = gw[a] - (fp.allocBand * nw);                                   // :632
            const float e = gw[a] + (fp.allocBand * nw);                              // :633
            dir[a] = e - start[a];
            cur[a] = voxel2block1(world2voxel1(start[a], fp.voxelSize));
            end[a] = voxel2block1(world2voxel1(e, fp.voxelSize));
        }
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            st[a] = dir[a] > 0.0f ? 1 : dir[a] < 0.0f ? -1 : 0;
            if (st[a] == 0) { tmax[a] = __builtin_inff(); tdelta[a] = __builtin_inff(); continue; }   // :658-668
            const float boundary = ((float)(int)((uint32_t)(cur[a] + (st[a] > 0 ? 1 : 0)) * 8u) - 0.5f) * fp.voxelSize;
            tmax[a] = (boundary - start[a]) / dir[a];
            tdelta[a] = (8.0f * fp.voxelSize) / __builtin_fabsf(dir[a]);
        }
        more = true;
    }

    // sample k of this lane's pixel: true if there is one (key in kx,ky,kz).  Must be called for k = 0, 1, 2, ...
    __device__ __forceinline__ bool key(const FrameParams &fp, const PixelVertex &p, int k, int &kx, int &ky, int &kz)
    {
        if (!p.valid) return false;
        if (dda) {
            if (k == 0) { kx = cur[0]; ky = cur[1]; kz = cur[2]; return true; }
            if (!more) return false;
            if (cur[0] == end[0] && cur[1] == end[1] && cur[2] == end[2]) { more = false; return false; }
            int a;
            if (tmax[0] < tmax[1] && tmax[0] < tmax[2]) a = 0;                        // :683
            else if (tmax[2] < tmax[1]) a = 2;                                        // :688
            else a = 1;                                                               // :693
            const float t = a == 0 ? tmax[0] : a == 1 ? tmax[1] : tmax[2];
            if (!(t <= 1.0f)) { more = false; return false; }
            if (a == 0) { cur[0] = (int)((uint32_t)cur[0] + (uint32_t)st[0]); tmax[0] += tdelta[0]; }
            else if (a == 1) { cur[1] = (int)((uint32_t)cur[1] + (uint32_t)st[1]); tmax[1] += tdelta[1]; }
            else { cur[2] = (int)((uint32_t)cur[2] + (uint32_t)st[2]); tmax[2] += tdelta[2]; }
            kx = cur[0]; ky = cur[1]; kz = cur[2];
            return true;
        }
        if (k >= nS) return false;
        const int half = (nS - 1) / 2;
        const float s = p.v.z + ((float)k - (float)half) * step;
        if (!(k == half || s > 0.0f)) return false;                      // the surface sample is never filtered (:621 only tests z != 0)
        float x = p.v.x, y = p.v.y, z = p.v.z;                           // k == half: the vertex itself, bit for bit
        if (k != half) {                                                 // wave-uniform; no divide on the reference path
            const float scale = (zOk && fast_range(s, 0x1p-50f, 0x1p50f)) ? div_fixed(s, p.v.z, zR1) : s / p.v.z;
            x = p.v.x * scale; y = p.v.y * scale; z = s;
        }
        const float4 g = mat4_mul(fp.T, x, y, z, p.v.w);                 // :622, w as stored
        int3_ b;
        if (nS > 1) {            // (wave-uniform)
            const FixedDivisor vs(fp.voxelSize, vsR1, vsOk);
            b = world2block(g.x, g.y, g.z, vs);                          // :636
        } else {
            b = world2block(g.x, g.y, g.z, fp.voxelSize);                // :636
        }
        kx = b.x; ky = b.y; kz = b.z;
        return true;
    }

    // wave-uniform: is there any lane that may still produce a sample with index >= k?
    __device__ __forceinline__ bool wave_done(int k) const
    {
        if (!dda) return k >= nS;
        return k >= kMaxBandSamples - 1 || (k > 0 && __ballot(more) == 0ull);
    }
};

// Sample k of this lane's pixel, frustum-tested and de-duplicated against the lane's own previous
// sample and against sample k of the lanes to the left and above (see the header comment).
template <bool kFrustumTest = true>
__device__ __forceinline__ SampleKey sample_key(const FrameParams &fp, const PixelVertex &p, BandWalk &walk, int k,
                                                int &ownX, int &ownY, int &ownZ, bool &ownHave)
{
    // The frustum test (:673) is a function of the key alone, so it runs after the dedup, on the few lanes
    // that survive it (a key that fails has no leader either way): per wave one or two evaluations instead
    // of 64 per sample.
    SampleKey r{0, 0, 0, false};
    const bool want = walk.key(fp, p, k, r.kx, r.ky, r.kz);
    const bool dupOwn = want && ownHave && ownX == r.kx && ownY == r.ky && ownZ == r.kz;
    if (want) { ownX = r.kx; ownY = r.ky; ownZ = r.kz; ownHave = true; }
    const int lane = threadIdx.x & (kWave - 1);
    const unsigned long long wants = __ballot(want);
    const int lx = __shfl_up(r.kx, 1), ly = __shfl_up(r.ky, 1), lz = __shfl_up(r.kz, 1);
    const int ux = __shfl_up(r.kx, 16), uy = __shfl_up(r.ky, 16), uz = __shfl_up(r.kz, 16);
    const bool dupLeft = (lane & 15) != 0 && ((wants >> (lane - 1)) & 1ull) && lx == r.kx && ly == r.ky && lz == r.kz;
    const bool dupUp = lane >= 16 && ((wants >> (lane - 16)) & 1ull) && ux == r.kx && uy == r.ky && uz == r.kz;
    r.leader = want && !dupOwn && !dupLeft && !dupUp;
    if (kFrustumTest && r.leader) r.leader = block_in_frustum(fp, r.kx, r.ky, r.kz);     // :673
    return r;
}

constexpr int kClaimQueue = 128;          // keys a wave queues before it probes them (band allocation)

__device__ __forceinline__ uint32_t sample_rank(const PixelVertex &p, int k)
{
    return (p.rank << kRankSampleBits) | (uint32_t)k;
}

// the claim phase for one 16x16 launch tile = one 256-lane workgroup (alloc_claim_kernel and the fused frame)
#ifdef VH_CLAIM_STAMPS
// diagnostics build (tools/ab_variants.sh build stamps -DVH_CLAIM_STAMPS; tools/claim_stamps.py): per claim tile the 100 MHz
// clock at its start, after the vertex load, after the sample loop and at its end, in the macro-cell bitmap's spare tail
__device__ unsigned long long *g_claimStamps = nullptr;
#define VH_CLAIM_STAMP(i) do { if (g_claimStamps && threadIdx.x == 0) g_claimStamps[(size_t)tile * 4 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define VH_CLAIM_STAMP(i) do { } while (0)
#endif
// Where a band-allocation tile's time goes, measured with per-tile time stamps (tools/claim_stamps.py) on C2 with a 10 cm
// band (5 samples per pixel): vertex load 2.3 us, the SAMPLE LOOP 8.5 us, drain of the key queue 1.8 us -- 12.6 us per
// tile, the last tile ending at 21.6 us of a 25 us launch whose walk needs 17.  The loop is instruction issue (~300
// instructions per sample and wave: three IEEE divisions, the frustum test of the leaders, six cross-lane moves), not a
// latency chain: splitting a tile's samples over two or three workgroups (shorter chains, the vertices read again) made
// the launch SLOWER (26.2 -> 29.2 -> 33.3 us), and so did fetching the bucket's first slot together with the pending
// frame's claim word in the drain (24.0 -> 26.1 us: more registers for every role of the fused kernel).
// kBand = false: the reference's frame only (one key per pixel; the caller guarantees fp.allocBand == 0): none of the band
// code is compiled in -- the pipelined kernel is 60 KB of instructions with it, about what a CU's instruction cache holds,
// and every role of the launch runs a different part of it at the same time
template <class In, bool kBand = true>
__device__ __forceinline__ void claim_tile(const FrameParams &fp, const DevPtrs &dp, const In &in, uint32_t tile,
                                           int candCounter, const Pending &pend = kNoPending,
                                           float *__restrict__ outDepth = nullptr, uint16_t *__restrict__ outRaw = nullptr,
                                           bool slotFirst = false)
{
    VH_CLAIM_STAMP(0);
    const PixelVertex p = load_pixel(fp, in, tile, threadIdx.x, outDepth, outRaw);
#ifdef VH_CLAIM_STAMPS
    if (p.v.z == 12345.678f) return;       // (forces the load to complete before the next stamp)
#endif
    VH_CLAIM_STAMP(1);
    if (!kBand) {
        // the surface sample alone: the vertex itself (:622), its block (:636), 2-D wave dedup, frustum test (:673), probe
        SampleKey s{0, 0, 0, false};
        if (p.valid) {
            const float4 g = mat4_mul(fp.T, p.v.x, p.v.y, p.v.z, p.v.w);
            const int3_ b = world2block(g.x, g.y, g.z, fp.voxelSize);
            s.kx = b.x; s.ky = b.y; s.kz = b.z;
        }
        const int ln = threadIdx.x & (kWave - 1);
        const unsigned long long wants = __ballot(p.valid);
        const int lx = __shfl_up(s.kx, 1), ly = __shfl_up(s.ky, 1), lz = __shfl_up(s.kz, 1);
        const int ux = __shfl_up(s.kx, 16), uy = __shfl_up(s.ky, 16), uz = __shfl_up(s.kz, 16);
        const bool dupLeft = (ln & 15) != 0 && ((wants >> (ln - 1)) & 1ull) && lx == s.kx && ly == s.ky && lz == s.kz;
        const bool dupUp = ln >= 16 && ((wants >> (ln - 16)) & 1ull) && ux == s.kx && uy == s.ky && uz == s.kz;
        if (!p.valid || dupLeft || dupUp) return;
        if ((fp.flags & kFlagWalkNt) || slotFirst) {
            // (slotFirst: the walk-free frame, where the claim tile's chain is the launch -- C2 walk-free 8.93 -> 8.83 us.)
            // A table beyond the Infinity Cache (the walk's loads are non-temporal then) answers the probes from HBM: the bucket's
            // first slot is requested BEFORE the frustum test (:673), which is a function of the key alone -- a key that fails it
            // has cost a read, every other key's chain is shorter by the test.  Same box, three rounds (profiles/r05_claim_early_slot_ab.txt):
            // C5table 3 468 -> 3 567 frames/s on average (noisy: +6 / -2 / +5 %), C3 +0.1 ... +1 %; the cache-resident C2 loses 0.7 %
            // (51.3 -> 50.95 k) and keeps the test first.
            const uint32_t h = hash_block(s.kx, s.ky, s.kz, fp.numBuckets);
            if (h < fp.bucketLo || h >= fp.bucketHi) return;            // not this shard's bucket
            const VoxelEntry first = dp.table[(size_t)(h - fp.bucketLo) * fp.bucketSize];
            if (!block_in_frustum(fp, s.kx, s.ky, s.kz)) return;
#ifdef VH_DEBUG_SKIP_ROLES
            if (fp.flags & kFlagDebugNoProbe) return;
#endif
            probe_and_claim<true>(fp, dp, s.kx, s.ky, s.kz, h, sample_rank(p, 0), candCounter, pend, (fp.flags & kFlagOverflow) ? nullptr : &first);
            return;
        }
        if (!block_in_frustum(fp, s.kx, s.ky, s.kz)) return;
        const uint32_t h = hash_block(s.kx, s.ky, s.kz, fp.numBuckets);
        if (h < fp.bucketLo || h >= fp.bucketHi) return;                // not this shard's bucket
#ifdef VH_DEBUG_SKIP_ROLES
        if (fp.flags & kFlagDebugNoProbe) return;
#endif
        probe_and_claim<false>(fp, dp, s.kx, s.ky, s.kz, h, sample_rank(p, 0), candCounter, pend);
        return;
    }
    BandWalk walk;
    walk.init(fp, p);
    int ox = 0, oy = 0, oz = 0;
    bool oh = false;
    if (walk.nS == 1 && !walk.dda) {
        // the reference's frame: one key per pixel, probed at once
        const SampleKey s = sample_key(fp, p, walk, 0, ox, oy, oz, oh);
        if (!s.leader) return;
        const uint32_t h = hash_block(s.kx, s.ky, s.kz, fp.numBuckets);
        if (h < fp.bucketLo || h >= fp.bucketHi) return;                // not this shard's bucket
        probe_and_claim(fp, dp, s.kx, s.ky, s.kz, h, sample_rank(p, 0), candCounter, pend);
        return;
    }
    // Band allocation: a pixel demands several keys.  Probing inside the sample loop puts one chain of
    // dependent bucket reads behind the other (5 samples: 5 round trips per wave); instead the wave queues
    // the keys that survive the dedup in LDS and probes them 64 at a time, one per lane, so the reads of a
    // whole queue are in flight together.  Who wins a bucket is decided by the ranks in the claim words, not
    // by the order of the probes.
    __shared__ int4 queues[256 / kWave][kClaimQueue];
    volatile int4 *queue = queues[threadIdx.x / kWave];
    const int lane = threadIdx.x & (kWave - 1);
    auto drain = [&](int count) {
        for (int base = 0; base < count; base += kWave) {
            if (base + lane < count) {
                const int kx = queue[base + lane].x, ky = queue[base + lane].y, kz = queue[base + lane].z;
                const uint32_t rank = (uint32_t)queue[base + lane].w;
                // the frustum test (:673) of the queued keys, one per lane: in the sample loop it ran for the one or two
                // leaders of a wave at a time
                if (block_in_frustum(fp, kx, ky, kz))
                    probe_and_claim<true>(fp, dp, kx, ky, kz, hash_block(kx, ky, kz, fp.numBuckets), rank, candCounter, pend);
            }
        }
    };
    int count = 0;                                                       // (wave-uniform)
    // (Measured in round 4 and not kept: for the DDA bands, every lane first taking 8 DDA steps on its own, then the dedup of
    // all 8 samples, then the queueing -- no cross-lane round trip between the steps.  Sample loop 6.16 -> 6.52 us per tile,
    // launch 22.2 -> 23.0 us: the loop is not a chain of cross-lane waits.)
    for (int k = 0; !walk.wave_done(k); ++k) {
        const SampleKey s = sample_key<false>(fp, p, walk, k, ox, oy, oz, oh);
        bool take = s.leader;
        if (take) {
            const uint32_t h = hash_block(s.kx, s.ky, s.kz, fp.numBuckets);
            take = h >= fp.bucketLo && h < fp.bucketHi;                  // this shard's bucket
        }
        const unsigned long long mask = __ballot(take);
        if (mask == 0ull) continue;
        const int n = __popcll(mask);
        if (count + n > kClaimQueue) {
            drain(count);
            count = 0;
        }
        if (take) {
            const int at = count + __popcll(mask & ((1ull << lane) - 1ull));
            queue[at].x = s.kx; queue[at].y = s.ky; queue[at].z = s.kz; queue[at].w = (int)sample_rank(p, k);
        }
        count += n;
    }
    VH_CLAIM_STAMP(2);
    drain(count);
    __builtin_amdgcn_s_waitcnt(0);
    VH_CLAIM_STAMP(3);
}

// The reference's frame (one key per pixel, no band) with a launch tile per WAVE instead of per workgroup: lane l takes pixels
// l, l + 64, l + 128, l + 192 of the tile -- the four 16x4 patches the four waves of claim_tile take -- keeps the same dedup per
// patch (so the same keys survive, with the same ranks), queues the survivors in LDS and probes them in one pass, one per lane.
// A quarter of the waves, each with four times the arithmetic behind the same round trips (vertex, bucket, claim word): the
// walk-free frame of a large image, whose launch is vector issue of thousands of claim tiles, gains (C3 28.4 -> 26.7 us); where
// the claim tile's chain is the launch's tail it loses -- C2 walk-free 8.9 -> 11.1 us, and under the reference's walk C2 18.9 ->
// 19.9 us at best (any claim_span) -- so the host selects it per frame (vh_api_frame.hip: claimPerWave).
constexpr int kWaveQueue = 64;            // survivors a wave queues before it probes them (a tile has ~9; a full queue is probed and refilled)
template <class In>
__device__ __forceinline__ void claim_tile_wave(const FrameParams &fp, const DevPtrs &dp, const In &in, uint32_t tile, int candCounter,
                                                const Pending &pend, float *__restrict__ outDepth, uint16_t *__restrict__ outRaw)
{
    __shared__ int4 waveKeys[256 / kWave][kWaveQueue];
    volatile int4 *queue = waveKeys[threadIdx.x / kWave];
    const int ln = threadIdx.x & (kWave - 1);
    const uint32_t tilesX = (uint32_t)(fp.width + 15) >> 4;
    const uint32_t by = tile / tilesX, bx = tile - by * tilesX;              // (wave-uniform)
    auto drain = [&](int n) {
        if (ln < n) {
            const int kx = queue[ln].x, ky = queue[ln].y, kz = queue[ln].z;
            const uint32_t rank = (uint32_t)queue[ln].w;
            const uint32_t h = hash_block(kx, ky, kz, fp.numBuckets);
            bool mine = h >= fp.bucketLo && h < fp.bucketHi;                 // this shard's bucket
#ifdef VH_DEBUG_SKIP_ROLES
            if (fp.flags & kFlagDebugNoProbe) mine = false;
#endif
            if (mine && (fp.flags & kFlagWalkNt)) {                          // (the first slot ahead of the frustum test, claim_tile: C3 walk-free 34.8 -> 35.5 k, C5table 36.5 -> 37.4 k)
                const VoxelEntry first = dp.table[(size_t)(h - fp.bucketLo) * fp.bucketSize];
                if (block_in_frustum(fp, kx, ky, kz))                        // :673
                    probe_and_claim<true>(fp, dp, kx, ky, kz, h, rank, candCounter, pend, (fp.flags & kFlagOverflow) ? nullptr : &first);
            } else if (mine && block_in_frustum(fp, kx, ky, kz)) {
                probe_and_claim<false>(fp, dp, kx, ky, kz, h, rank, candCounter, pend);
            }
        }
    };
    PixelVertex p[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) p[j] = load_pixel_at(fp, in, tile, bx, by, (uint32_t)ln + 64u * (uint32_t)j, outDepth, outRaw);
    int count = 0;                                                           // (wave-uniform)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int kx = 0, ky = 0, kz = 0;
        if (p[j].valid) {
            const float4 g = mat4_mul(fp.T, p[j].v.x, p[j].v.y, p[j].v.z, p[j].v.w);       // :622
            const int3_ b = world2block(g.x, g.y, g.z, fp.voxelSize);                       // :636
            kx = b.x; ky = b.y; kz = b.z;
        }
        const unsigned long long wants = __ballot(p[j].valid);
        const int lx = __shfl_up(kx, 1), ly = __shfl_up(ky, 1), lz = __shfl_up(kz, 1);
        const int ux = __shfl_up(kx, 16), uy = __shfl_up(ky, 16), uz = __shfl_up(kz, 16);
        const bool dupLeft = (ln & 15) != 0 && ((wants >> (ln - 1)) & 1ull) && lx == kx && ly == ky && lz == kz;
        const bool dupUp = ln >= 16 && ((wants >> (ln - 16)) & 1ull) && ux == kx && uy == ky && uz == kz;
        const bool leader = p[j].valid && !dupLeft && !dupUp;
        const unsigned long long mask = __ballot(leader);
        if (mask == 0ull) continue;
        const int n = __popcll(mask);
        if (count + n > kWaveQueue) {
            drain(count);
            count = 0;
        }
        if (leader) {
            const int at = count + __popcll(mask & ((1ull << ln) - 1ull));
            queue[at].x = kx; queue[at].y = ky; queue[at].z = kz; queue[at].w = (int)sample_rank(p[j], 0);
        }
        count += n;
    }
    drain(count);
}

template <class In>
__global__ __launch_bounds__(256) void alloc_claim_kernel(const FrameParams fp, const DevPtrs dp, const In in)
{
    claim_tile(fp, dp, in, blockIdx.x, kCandCount);
}

// Key generation for the multi-GPU exchange: the same per-pixel
// work, but the surviving keys are binned by owning shard instead of probed.  Slots in
// a bin come from one global counter per bin; to keep that word off the critical path
// (one address sustains only ~90 returning atomics per microsecond) a 1024-lane
// workgroup (four launch tiles) first counts its keys per owner in LDS and then takes one
// global atomicAdd per owner it actually has keys for.
#ifndef VH_GEN_THREADS
#define VH_GEN_THREADS 1024
#endif
constexpr int kGenThreads = VH_GEN_THREADS;   // tuning knob (make EXTRA=-DVH_GEN_THREADS=n); a multiple of 256
// The batched kernels choose between this and half of it by the number of owners: what bounds them is the returning atomic on
// the bin headers (~90 per microsecond and address).  One owner: every workgroup of the batch hits ONE word, so fewer, larger
// workgroups win (8 frames of 640x480: 38.9 us with 1024 lanes, 60 with 512, 110 with 256); eight owners: the words share the
// load and shorter chains win (38.4 / 28.3 / 33.6 us).
constexpr int kGenTiles = kGenThreads / 256;

template <class In, int kThreads = kGenThreads>
__device__ __forceinline__ void generate_keys_tile(const FrameParams &fp, const In &verts,
                                                   int32_t numShards, int4 *__restrict__ outBins,
                                                   int32_t outCapacity, int32_t outBinStride,
                                                   float *__restrict__ outDepth, uint32_t rankBase, uint32_t group)
{
    __shared__ int ldsCount[VH_MAX_CAMERAS];
    __shared__ int ldsBase[VH_MAX_CAMERAS];
    const PixelVertex p = load_pixel(fp, verts, group * (kThreads / 256) + (threadIdx.x >> 8), threadIdx.x & 255u, outDepth);
    BandWalk walk;
    walk.init(fp, p);
    const uint32_t perShard = (fp.numBuckets + (uint32_t)numShards - 1u) / (uint32_t)numShards;
    int ox = 0, oy = 0, oz = 0;
    bool oh = false;
    // (the ray band's sample count is the same for every lane of the workgroup; the ray DDA runs until no lane of the
    // workgroup has a step left; the normal DDA is not offered on this path: vh_generate_keys* carry no normal map)
    for (int k = 0; k < kMaxBandSamples - 1; ++k) {
        if (walk.dda ? (k > 0 && !__syncthreads_or(walk.more ? 1 : 0)) : k >= walk.nS) break;
        if (threadIdx.x < VH_MAX_CAMERAS) ldsCount[threadIdx.x] = 0;
        __syncthreads();
        const SampleKey s = sample_key(fp, p, walk, k, ox, oy, oz, oh);
        uint32_t owner = 0;
        int local = 0;
        if (s.leader) {
            owner = hash_block(s.kx, s.ky, s.kz, fp.numBuckets) / perShard;
            local = atomicAdd(&ldsCount[owner], 1);
        }
        __syncthreads();
        if ((int)threadIdx.x < numShards && ldsCount[threadIdx.x] > 0)
            ldsBase[threadIdx.x] = atomicAdd(&outBins[(size_t)threadIdx.x * outBinStride].x, ldsCount[threadIdx.x]);
        __syncthreads();
        if (s.leader) {
            int4 *bin = outBins + (size_t)owner * outBinStride;           // record 0 = {count,0,0,0}
            const int slot = ldsBase[owner] + local + 1;
            if (slot < outCapacity) bin[slot] = make_int4(s.kx, s.ky, s.kz, (int)(rankBase + sample_rank(p, k)));
        }
        __syncthreads();
    }
}

// The reference's frame (one key per pixel, fp.allocBand == 0) with kGroups pixel groups per workgroup: the keys of all groups
// are counted in LDS first and the workgroup takes ONE global atomicAdd per owner for all of them -- the returning atomics on
// the bin headers are what bounds the launch (above), and this divides their number by kGroups.
// frameSlot >= 0 (fused generation, vh_shard.hip): the frames of a batch are generated by one launch after the other, and each
// counts its records in a counter of its OWN -- int frameSlot of the bin's last two records (outCapacity excludes them) -- and writes
// them behind the records of the frames before it, whose counters are final: a frame's records are contiguous, the consumer finds
// them by a prefix sum of the eight counters, and nobody has to mark the end of a frame.
template <class In, int kThreads, int kGroups>
__device__ __forceinline__ void generate_keys_groups(const FrameParams &fp, const In &verts, int32_t numShards,
                                                     int4 *__restrict__ outBins, int32_t outCapacity, int32_t outBinStride,
                                                     float *__restrict__ outDepth, uint32_t rankBase, uint32_t firstGroup, int frameSlot = -1)
{
    __shared__ int ldsCount[VH_MAX_CAMERAS];
    __shared__ int ldsBase[VH_MAX_CAMERAS];
    if (threadIdx.x < VH_MAX_CAMERAS) ldsCount[threadIdx.x] = 0;
    int frameStart = 0;                       // (lane = owner) records of the frames before this one
    if (frameSlot >= 0 && (int)threadIdx.x < numShards) {
        const int *cnt = reinterpret_cast<const int *>(outBins + (size_t)threadIdx.x * outBinStride + outCapacity);
        for (int f = 0; f < frameSlot; ++f) frameStart += cnt[f];
    }
    __syncthreads();
    const uint32_t perShard = (fp.numBuckets + (uint32_t)numShards - 1u) / (uint32_t)numShards;
    const int ln = threadIdx.x & (kWave - 1);
    int4 rec[kGroups];
    int where[kGroups];                       // owner << 20 | index among the workgroup's keys for that owner; -1: no key
    // (the workgroup's tiles are consecutive: one division for the first, the rest by stepping -- wave-uniform)
    const uint32_t tilesX = (uint32_t)(fp.width + 15) >> 4;
    uint32_t tile = firstGroup * (kThreads / 256) + (threadIdx.x >> 8);
    uint32_t tby = tile / tilesX, tbx = tile - tby * tilesX;
#pragma unroll
    for (int g = 0; g < kGroups; ++g) {
        const PixelVertex p = load_pixel_at(fp, verts, tile, tbx, tby, threadIdx.x & 255u, outDepth);
        tile += kThreads / 256;
        tbx += kThreads / 256;
        while (tbx >= tilesX) { tbx -= tilesX; ++tby; }
        int kx = 0, ky = 0, kz = 0;
        if (p.valid) {
            const float4 w = mat4_mul(fp.T, p.v.x, p.v.y, p.v.z, p.v.w);               // :622
            const int3_ b = world2block(w.x, w.y, w.z, fp.voxelSize);                   // :636 (div_fixed with the reciprocal shared by the
                                                                                        //  workgroup's groups, the same bits: 49.3 -> 46.8 k frames/s)
            kx = b.x; ky = b.y; kz = b.z;
        }
        const unsigned long long wants = __ballot(p.valid);
        const int lx = __shfl_up(kx, 1), ly = __shfl_up(ky, 1), lz = __shfl_up(kz, 1);
        const int ux = __shfl_up(kx, 16), uy = __shfl_up(ky, 16), uz = __shfl_up(kz, 16);
        const bool dupLeft = (ln & 15) != 0 && ((wants >> (ln - 1)) & 1ull) && lx == kx && ly == ky && lz == kz;
        const bool dupUp = ln >= 16 && ((wants >> (ln - 16)) & 1ull) && ux == kx && uy == ky && uz == kz;
        where[g] = -1;
        if (p.valid && !dupLeft && !dupUp && block_in_frustum(fp, kx, ky, kz)) {          // :673
            // (wave-uniform branches: one owner has nothing to divide; a power-of-two range per owner -- every BASELINE config -- shifts.
            //  The generating workgroups are the longest chain of the fused launch: 47.8 -> 49.0 k frames/s with one rank)
            const uint32_t owner = numShards == 1 ? 0u
                                 : (perShard & (perShard - 1u)) == 0u ? hash_block(kx, ky, kz, fp.numBuckets) >> (31 - __builtin_clz(perShard))
                                                                      : hash_block(kx, ky, kz, fp.numBuckets) / perShard;
            where[g] = (int)(owner << 20) | atomicAdd(&ldsCount[owner], 1);
            rec[g] = make_int4(kx, ky, kz, (int)(rankBase + sample_rank(p, 0)));
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < numShards && ldsCount[threadIdx.x] > 0) {
        if (frameSlot >= 0)
            ldsBase[threadIdx.x] = frameStart + atomicAdd(reinterpret_cast<int *>(outBins + (size_t)threadIdx.x * outBinStride + outCapacity) + frameSlot,
                                                          ldsCount[threadIdx.x]);
        else
            ldsBase[threadIdx.x] = atomicAdd(&outBins[(size_t)threadIdx.x * outBinStride].x, ldsCount[threadIdx.x]);
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < kGroups; ++g) {
        if (where[g] >= 0) {
            const uint32_t owner = (uint32_t)where[g] >> 20;
            const int slot = ldsBase[owner] + (where[g] & 0xfffff) + 1;                    // record 0 = {count,0,0,0}
            if (slot < outCapacity) outBins[(size_t)owner * outBinStride + slot] = rec[g];
        }
    }
}

__global__ __launch_bounds__(kGenThreads) void generate_keys_kernel(const FrameParams fp,
                                                                    const float4 *__restrict__ verts,
                                                                    int32_t numShards, int4 *__restrict__ outBins,
                                                                    int32_t outCapacity, int32_t outBinStride,
                                                                    float *__restrict__ outDepth, uint32_t rankBase)
{
    if (outDepth && blockIdx.x == 0 && threadIdx.x < kPacketHeader)      // packet header: pose, inverse
        outDepth[(int)threadIdx.x - kPacketHeader] = threadIdx.x < 16 ? fp.T[threadIdx.x] : fp.Tinv[threadIdx.x - 16];
    generate_keys_tile(fp, VertexMap{verts, nullptr}, numShards, outBins, outCapacity, outBinStride, outDepth, rankBase,
                       blockIdx.x);
}

// Up to kGenBatch frames of one camera in ONE launch (blockIdx.y = frame): a single frame is 300
// latency-bound workgroups, which leave most of the chip idle; the frames of an exchange batch run
// side by side instead of one after the other.  Poses and vertex-map pointers travel in the kernel
// arguments.
constexpr int kGenBatch = 8;
struct GenFrames {
    float T[kGenBatch][16];
    float Tinv[kGenBatch][16];
    const float4 *verts[kGenBatch];
};

template <int kThreads>
__global__ __launch_bounds__(kThreads) void generate_keys_batch_kernel(FrameParams fp, const GenFrames fr,
                                                                          int32_t numShards,
                                                                          int4 *__restrict__ outBins,
                                                                          int32_t outCapacity, int32_t outBinStride,
                                                                          int32_t frameStride,
                                                                          float *__restrict__ packets,
                                                                          size_t packetFrameStride, uint32_t rankBase)
{
    const int b = blockIdx.y;
    float *outDepth = packets ? packets + packetFrameStride * b + kPacketHeader : nullptr;
    // (the header is written from the argument block: indexing the private copy of fp by lane
    // would push it to scratch memory)
    if (outDepth && blockIdx.x == 0 && threadIdx.x < kPacketHeader)
        outDepth[(int)threadIdx.x - kPacketHeader] = threadIdx.x < 16 ? fr.T[b][threadIdx.x] : fr.Tinv[b][threadIdx.x - 16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { fp.T[i] = fr.T[b][i]; fp.Tinv[i] = fr.Tinv[b][i]; }
    // (frameStride < 0: one bin per owner for the whole batch; rankBase then carries the frame index where the camera id sits)
    generate_keys_tile<VertexMap, kThreads>(fp, VertexMap{fr.verts[b], nullptr}, numShards, frameStride < 0 ? outBins : outBins + (size_t)frameStride * b,
                       outCapacity, outBinStride, outDepth,
                       frameStride < 0 ? rankBase + ((uint32_t)b << kRankCameraShift) : rankBase, blockIdx.x);
}

// The same from uint16 sensor images: vertices computed in place (SensorImage), and the packet of
// a frame is written by the same launch -- header {pose, inverse, K_inv row 2, unit} and the image
// itself (VH_PACKET_U16) -- so a whole exchange batch of one camera is ONE launch.
struct GenSensorFrames {
    float T[kGenBatch][16];
    float Tinv[kGenBatch][16];
    const uint16_t *depth[kGenBatch];
    float k[9];
    float unit;
};

template <int kThreads, int kGroups = 0>        // kGroups > 0: no band (the caller checks), kGroups pixel groups per workgroup
__global__ __launch_bounds__(kThreads) void generate_keys_sensor_batch_kernel(FrameParams fp, const GenSensorFrames fr,
                                                                                 int32_t numShards,
                                                                                 int4 *__restrict__ outBins,
                                                                                 int32_t outCapacity,
                                                                                 int32_t outBinStride,
                                                                                 int32_t frameStride,
                                                                                 float *__restrict__ packets,
                                                                                 size_t packetFrameStride,
                                                                                 uint32_t rankBase)
{
    const int b = blockIdx.y;
    SensorImage in;
    in.depth = fr.depth[b];
#pragma unroll
    for (int i = 0; i < 9; ++i) in.k[i] = fr.k[i];
    in.unit = fr.unit;
    if (packets) {
        float *pk = packets + packetFrameStride * b;
        if (blockIdx.x == 0 && threadIdx.x < kPacketHeaderU16) {
            const int t = threadIdx.x;
            pk[t] = t < 16 ? fr.T[b][t] : t < 32 ? fr.Tinv[b][t - 16] : t == 32 ? fr.k[6] : t == 33 ? fr.k[7]
                                                                                  : t == 34 ? fr.k[8] : fr.unit;
        }
        // straight copy of the image, one pixel per lane (the grid covers ceil(tiles/4)*1024 >= W*H lanes)
#pragma unroll
        for (int g = 0; g < (kGroups > 0 ? kGroups : 1); ++g) {
            const int idx = (blockIdx.x * (kGroups > 0 ? kGroups : 1) + g) * kThreads + threadIdx.x;
            if (idx < fp.width * fp.height) reinterpret_cast<uint16_t *>(pk + kPacketHeaderU16)[idx] = in.depth[idx];
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) { fp.T[i] = fr.T[b][i]; fp.Tinv[i] = fr.Tinv[b][i]; }
    int4 *bins = frameStride < 0 ? outBins : outBins + (size_t)frameStride * b;
    const uint32_t base = frameStride < 0 ? rankBase + ((uint32_t)b << kRankCameraShift) : rankBase;
    if (kGroups > 0) generate_keys_groups<SensorImage, kThreads, (kGroups > 0 ? kGroups : 1)>(fp, in, numShards, bins, outCapacity, outBinStride, nullptr, base, blockIdx.x * (uint32_t)kGroups);
    else generate_keys_tile<SensorImage, kThreads>(fp, in, numShards, bins, outCapacity, outBinStride, nullptr, base, blockIdx.x);
}

// ---------------------------------------------------------------------------
// allocBlocks, phase 2
// ---------------------------------------------------------------------------
// Exactly one contender per bucket finds its own word in the claim array: the
// one with the lowest launch rank, i.e. the thread a sequential run of the
// reference grid would have let through the atomicExch (VoxelUtils.cu:444-445).
// It takes the first free slot and pops the heap (top-down, :328-334).  An empty
// heap refuses the insertion instead of reading heap[-1].
// Raycast accelerator: "macro cells" of 4x4x4 blocks, one bit per hashed macro coordinate
// (collisions only make the ray skip less).  Set when a block inside the cell is inserted.
constexpr uint32_t kMacroBits = 1u << 20;      // 128 KB bitmap

__device__ __forceinline__ uint32_t macro_hash(int mx, int my, int mz)
{
    return (((uint32_t)mx * 73856093u) ^ ((uint32_t)my * 19349669u) ^ ((uint32_t)mz * 83492791u)) & (kMacroBits - 1u);
}

// Returns true (and the new entry) if candidate k held its bucket's claim and was inserted.
// `index`: position of the candidate in dp.candidates (its look-ahead target sits beside it)
// consume = false (pipelined frames): the claim words stay as they are -- the claim phase and the
// walk of the NEXT frame read them while this commit runs; nobody stakes a claim in this epoch any more.
// epoch / claim: the lock epoch and claim array of the frame being committed (fp.epoch / dp.claim; a pipelined multi-camera
// launch passes the previous frame's)
__device__ __forceinline__ bool commit_candidate(const FrameParams &fp, const DevPtrs &dp, const int4 k,
                                                 VoxelEntry &e, uint32_t index, bool consume, uint32_t epoch,
                                                 unsigned long long *__restrict__ claim)
{
    const uint32_t h = hash_block(k.x, k.y, k.z, fp.numBuckets);
    const uint32_t local = h - fp.bucketLo;
    const unsigned long long w = claim[local];
    if (claim_epoch(w) != epoch || claim_slot(w) != index || w == consumed_word(epoch)) return false;   // lost the bucket this frame
    if (consume) claim[local] = consumed_word(epoch);                            // locked until the next epoch
    const uint32_t target = (fp.flags & kFlagOverflow) ? dp.candTarget[index] : ~0u;
    if (target != ~0u) {
        // home bucket full: the entry goes to the free slot found behind it and to the FRONT of the
        // bucket's chain -- if this contender also holds the bucket of that slot
        const uint32_t tb = target / fp.bucketSize;
        const unsigned long long wt = claim[tb];
        if (claim_epoch(wt) != epoch || claim_slot(wt) != index || wt == consumed_word(epoch))
            return false;                                                        // the home bucket stays locked, as in the reference
        claim[tb] = consumed_word(epoch);
        const int addr = atomicSub(dp.counters + kHeapCounter, 1);
        if (addr < 0) {
            atomicAdd(dp.counters + kHeapCounter, 1);
            atomicAdd(dp.counters + kHeapExhausted, 1);
            return false;
        }
        const uint32_t last = local * fp.bucketSize + fp.bucketSize - 1u, n = owned_entries(fp);
        e.pos[0] = k.x; e.pos[1] = k.y; e.pos[2] = k.z;
        e.ptr = (int)(dp.heap[addr] * (uint32_t)kBlockVoxels);
        e.offset = dp.table[last].offset;
        dp.table[target] = e;
        dp.table[last].offset = (int)(target >= last ? target - last : target + n - last);
        atomicOr(dp.bucketBits + (tb >> 5), 1u << (tb & 31u));
        const uint32_t hm = macro_hash(k.x >> 2, k.y >> 2, k.z >> 2);
        atomicOr(dp.macroBits + (hm >> 5), 1u << (hm & 31u));
        atomicAdd(dp.counters + kAllocatedTotal, 1);
        return true;
    }
    VoxelEntry *bucket = dp.table + (size_t)local * fp.bucketSize;
    for (uint32_t s = 0; s < fp.bucketSize; ++s) {
        if (bucket[s].ptr != VH_FREE_BLOCK) continue;
        const int addr = atomicSub(dp.counters + kHeapCounter, 1);
        if (addr < 0) {                                   // heap empty: undo, refuse
            atomicAdd(dp.counters + kHeapCounter, 1);
            atomicAdd(dp.counters + kHeapExhausted, 1);
            return false;
        }
        e.pos[0] = k.x; e.pos[1] = k.y; e.pos[2] = k.z;
        e.ptr = (int)(dp.heap[addr] * (uint32_t)kBlockVoxels);
        e.offset = 0;
        bucket[s] = e;
        atomicOr(dp.bucketBits + (local >> 5), 1u << (local & 31u));
        const uint32_t hm = macro_hash(k.x >> 2, k.y >> 2, k.z >> 2);
        atomicOr(dp.macroBits + (hm >> 5), 1u << (hm & 31u));
        atomicAdd(dp.counters + kAllocatedTotal, 1);
        return true;
    }
    return false;
}

__device__ __forceinline__ bool commit_candidate(const FrameParams &fp, const DevPtrs &dp, const int4 k,
                                                 VoxelEntry &e, uint32_t index, bool consume = true)
{
    return commit_candidate(fp, dp, k, e, index, consume, fp.epoch, dp.claim);
}

__global__ __launch_bounds__(256) void alloc_commit_kernel(const FrameParams fp, const DevPtrs dp)
{
    int n = dp.counters[kCandCount];
    if ((uint32_t)n > dp.candCapacity) n = (int)dp.candCapacity;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        VoxelEntry e;
        (void)commit_candidate(fp, dp, dp.candidates[i], e, (uint32_t)i);
    }
    // the last workgroup to finish re-arms the per-frame counters
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const int ticket = atomicAdd(dp.counters + kCommitTicket, 1);
        if (ticket == (int)gridDim.x - 1) {
            dp.counters[kLastCandidates] = dp.counters[kCandCount];
            dp.counters[kCandCount] = 0;
            dp.counters[kCompactCount] = 0;
            dp.counters[kCommitTicket] = 0;
        }
    }
}

}  // namespace vh
