// vh_raycast_coop.hip -- the cooperative DDA raycast: one block list per wave (an 8x8 ray patch).
// Part of libvoxelhash_hip.so (gfx950); included after vh_raycast.hip (DDA helpers, beam front end, per-lane walk).
#pragma once

namespace vh {

// ---------------------------------------------------------------------------
// The cooperative form (RaycastArgs::beam == 2, the default): one block list per wave
// ---------------------------------------------------------------------------
// The per-lane walk above spends its time where the 64 rays of a wave do the same thing 64 times, out of step with
// each other: stepping through the absent blocks in front of the surface, looking buckets up, rebuilding voxel
// coordinates -- with a dependent gather (64 different cache lines, ~500 cycles) at every step, and with the rare
// path (a bucket bit is set) entered in most rounds because SOME lane needs it (per-wave timeline on C2: 10-16 loop
// rounds of 1.6 us each; the launch is as long as its slowest wave, 60-70 us, the fixed-step march 47).  The rays
// of a patch are a few voxels apart, though, and meet the same handful of blocks.  So the wave finds those blocks
// ONCE, together, and every ray is then tested against each of them directly:
//   1. beam: lane i bounds the part of the beam inside half-block slab i of the depth range by a box in voxel-grid
//      units and tests the (at most 2 x 2 x 2) blocks the box touches: eight independent bucket-bit loads, one round
//      trip for all 64 slabs.  Cells with a set bit go into a wave-local set in LDS (compare-and-swap on a 32-bit tag
//      that IS the key: block coordinates relative to the wave's first block, 10 bits each);
//   2. the set's cells are resolved against the hash table, four per lane, their bucket's first entry fetched
//      together (getVoxelEntry4Block, VoxelUtils.cu:362-382): the allocated ones form the wave's block list;
//   3. for every block of the list (a wave-uniform loop: its key and its voxel pointer are scalars) each lane
//      computes, in the walk's own arithmetic, whether and where its ray enters the block -- the ray is inside the
//      block's slab on axis a between the event that steps c_a into it and the event that steps c_a out; it visits
//      the block iff the last of the three entering events precedes the first of the three leaving events in the
//      merge order; the voxel it enters at follows from dda_advance as in the per-lane walk -- and walks its voxels.
// The blocks are judged independently, in whatever order the list has: a pair of consecutive valid samples lies
// inside one block, or its first sample is the voxel the ray was in before the block's entry event -- looked up
// through the same set -- so every block yields its candidate hits without knowing what came before, each candidate
// carries the event at which the ray arrived in its voxel, and the ray's hit is the candidate with the earliest
// arrival (events are totally ordered).  Complete by construction: the slabs cover [t_min, t_max] (several windows
// of 64 when the range is longer), the boxes are conservative, so every allocated block any ray of the wave visits
// is in the list.  Whenever the preconditions fail -- a box spans more than two blocks on an axis, the set
// overflows, a block lies more than 511 blocks from the wave's first -- the wave falls back to the per-lane walk.
constexpr int kCoopSubs = 4;                          // slabs per lane and window of step 1 (64 x this many half-block slabs)
constexpr int kCoopSlots = 256;                       // per wave: cells with a set bucket bit
constexpr uint32_t kCoopUnresolved = 0x7ffffffeu;     // sPtr: not looked up yet
struct CoopShared {
    uint32_t tag[kDdaBlockWaves][kCoopSlots];
    uint32_t ptr[kDdaBlockWaves][kCoopSlots];
    uint32_t list[kDdaBlockWaves][kCoopSlots];     // allocated cells: slot | depth key << 16 (the list is walked front to back)
    uint16_t cells[kDdaBlockWaves][kCoopSlots];    // the set's occupied slots in order of insertion (what step 2 resolves)
    uint32_t count[kDdaBlockWaves];
};

__device__ __forceinline__ uint32_t coop_tag(int rx, int ry, int rz) { return 1u + (uint32_t)rx + ((uint32_t)ry << 10) + ((uint32_t)rz << 20); }

// slot of `tag` in the wave's set, or -1
__device__ __forceinline__ int coop_find(const uint32_t *tags, uint32_t tag)
{
    uint32_t h = (tag * 2654435761u) >> 24;
    for (int probe = 0; probe < 16; ++probe) {
        const uint32_t t = tags[h];
        if (t == tag) return (int)h;
        if (t == 0u) return -1;
        h = (h + 1u) & (kCoopSlots - 1);
    }
    return -1;
}

// Whether and where a ray enters block kk: the ray is inside the block's slab on axis a from the event that steps c_a
// into it to the event that steps c_a out of it, and it visits the block iff the LAST of the three entering events
// precedes the FIRST of the three leaving events in the merge order (or is the same event).
struct CoopEntry {
    float tE;              // the entering event (-inf: the ray starts inside the block)
    int pE, xe;            // its priority and axis
    bool inside, enters;
    bool startIn[3];       // the ray starts inside the slab of axis a
};
__device__ __forceinline__ CoopEntry coop_entry(const DdaAxis (&ax)[3], const int (&c)[3], const int (&kk)[3], float tMax)
{
    CoopEntry r;
    float tIn[3], tOut[3];
    bool miss = false;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int lo = kk[a] << 3, hi = lo + 7;
        if (ax[a].invE == 0.0f) {
            r.startIn[a] = c[a] >= lo && c[a] <= hi;
            miss |= !r.startIn[a];
            tIn[a] = -__builtin_inff(); tOut[a] = __builtin_inff();
        } else if (ax[a].s > 0) {
            miss |= c[a] > hi;
            r.startIn[a] = c[a] >= lo;
            tIn[a] = r.startIn[a] ? -__builtin_inff() : dda_tnext(ax[a], lo - 1);
            tOut[a] = dda_tnext(ax[a], hi);
        } else {
            miss |= c[a] < lo;
            r.startIn[a] = c[a] <= hi;
            tIn[a] = r.startIn[a] ? -__builtin_inff() : dda_tnext(ax[a], hi + 1);
            tOut[a] = dda_tnext(ax[a], lo);
        }
    }
    // the LAST entering event and the FIRST leaving event in merge order
    int xe = dda_before(tIn[0], 2, tIn[1], 0) ? 1 : 0;
    {
        const float t01 = xe ? tIn[1] : tIn[0];
        if (dda_before(t01, xe ? 0 : 2, tIn[2], 1)) xe = 2;
    }
    const int xo = (tOut[0] < tOut[1] && tOut[0] < tOut[2]) ? 0 : (tOut[2] < tOut[1]) ? 2 : 1;
    r.tE = xe == 0 ? tIn[0] : xe == 1 ? tIn[1] : tIn[2];
    const float tO = xo == 0 ? tOut[0] : xo == 1 ? tOut[1] : tOut[2];
    r.pE = xe == 0 ? 2 : xe == 1 ? 0 : 1;
    const int pO = xo == 0 ? 2 : xo == 1 ? 0 : 1;
    r.xe = xe;
    r.inside = r.tE == -__builtin_inff();
    r.enters = !miss && (r.inside || xe == xo || dda_before(r.tE, r.pE, tO, pO)) && (r.inside || r.tE < tMax);
    return r;
}

constexpr int kCoopK = 1;            // voxels fetched per round trip in the block walk (2 and 3: slower in rounds 4 and 6, DESIGN_LOG.md)
template <bool kNormals>
__global__ __launch_bounds__(64 * kDdaBlockWaves, VH_DDA_WAVES) void raycast_coop_kernel(const FrameParams fp, const DevPtrs dp, const RaycastArgs ra,
                                                          float *__restrict__ depthOut, float4 *__restrict__ normalOut)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long stamp0 = ra.stamps ? __builtin_amdgcn_s_memrealtime() : 0ull;
    int tx = blockIdx.x, ty = blockIdx.y;
    xcd_tile(tx, ty);                        // each XCD (own L2) renders a contiguous run of image tiles
    // the wave's pixel patch: 8x8 of the workgroup's 16x16 tile
    const int pu = tx * 16 + (wave & 1) * 8, pv = ty * 16 + (wave >> 1) * 8;
    const int u = pu + (lane & 7), v = pv + (lane >> 3);
    const bool inImage = u < fp.width && v < fp.height;
    DdaAxis ax[3];
    int c[3];
    float dx, dy;
    dda_ray(fp, ra, u, v, ax, c, dx, dy);
    const float vs = fp.voxelSize;
    bool live = inImage;
    bool found = false;
    float hit = 0.0f;
    int hx = 0, hy = 0, hz = 0, hptr = VH_FREE_BLOCK;          // (per-lane walk: the last valid sample's voxel;) after a hit: the hit voxel and its block
    const int prio[3] = {2, 0, 1};
    bool coopDone = false;
    unsigned long long stampP1 = 0ull;                        // diagnostics: the ray set-up is done
    unsigned long long stampA = stamp0, stampB = stamp0;      // diagnostics: the set is built / the list is resolved
    int coopList = 0, coopWalks = 0;
    {
        __shared__ CoopShared sh_;
        uint32_t *tags = sh_.tag[wave], *ptrs = sh_.ptr[wave];
        uint32_t *list = sh_.list[wave];
#pragma unroll
        for (int r = 0; r < kCoopSlots / 64; ++r) tags[lane + 64 * r] = 0u;
        uint16_t *cells = sh_.cells[wave];
        uint32_t *count = &sh_.count[wave];
        if (lane == 0) *count = 0u;
        int nCells = 0;
        float eMin[3], eMax[3];                                    // wave-uniform: the patch's corner rays
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float e0 = __shfl(ax[a].E, 0), e1 = __shfl(ax[a].E, 7);
            const float e2 = __shfl(ax[a].E, 56), e3 = __shfl(ax[a].E, 63);
            eMin[a] = __builtin_fminf(__builtin_fminf(e0, e1), __builtin_fminf(e2, e3));
            eMax[a] = __builtin_fmaxf(__builtin_fmaxf(e0, e1), __builtin_fmaxf(e2, e3));
        }
        const int base0 = (__shfl(c[0], 0) >> 3) - 512, base1 = (__shfl(c[1], 0) >> 3) - 512, base2 = (__shfl(c[2], 0) >> 3) - 512;
        const float dt2 = 4.0f * vs;                               // half-block slabs
        // the walk's per-ray constants
        const float f0 = (float)ax[0].s, f1 = (float)ax[1].s, f2 = (float)ax[2].s;
        // (an axis that never steps: invE = 0 would give a crossing time of 0; (c + 1e30) * inf = inf instead)
        const float ie0 = ax[0].invE != 0.0f ? ax[0].invE : __builtin_inff(), gs0 = ax[0].invE != 0.0f ? ax[0].Gs : -1.0e30f;
        const float ie1 = ax[1].invE != 0.0f ? ax[1].invE : __builtin_inff(), gs1 = ax[1].invE != 0.0f ? ax[1].Gs : -1.0e30f;
        const float ie2 = ax[2].invE != 0.0f ? ax[2].invE : __builtin_inff(), gs2 = ax[2].invE != 0.0f ? ax[2].Gs : -1.0e30f;
        const int d0 = ax[0].s * ((1 << 16) + 1), d1 = ax[1].s * ((8 << 16) + (1 << 5)), d2 = ax[2].s * ((64 << 16) + (1 << 10));
        bool fail = false;
        float bestT = __builtin_inff();                            // arrival event of the best candidate's hit voxel
        int bestP = 3;
        int recW = -1;                                             // where the best candidate's pair sits (-1: none yet)
        float recPs = 0.0f, recSdf = 0.0f;                         // its two samples
        int nList = 0;
        bool final_ = !inImage;
        __builtin_amdgcn_wave_barrier();
        stampP1 = ra.stamps ? __builtin_amdgcn_s_memrealtime() : 0ull;
        // A window = the slabs one pass of step 1 covers: 64 per sub-pass, up to kCoopSubs sub-passes when the range is longer
        // (finer voxels).  One window for the whole range beats several (640x480, 60 frames, per call: 2 windows of 64 slabs
        // 65.5 us, 3: 83.6, 4: 102.0 -- every window pays the front end again, and the per-lane walk behind a beam front end,
        // 58.9 / 74.3 / 92.5, was faster); what lies behind a ray's hit is then listed too, but skipped by its arrival event.
        const int nSub = max(1, min(kCoopSubs, (int)__builtin_ceilf((ra.tMax - ra.tMin) / (64.0f * dt2))));
        const float window = 64.0f * (float)nSub * dt2;
        for (float tw = ra.tMin; tw < ra.tMax && !fail; tw += window) {
            // ---- 1. beam: the blocks slab `lane` (+ 64 per sub-pass) of this window can touch ----
            for (int sub = 0; sub < nSub; ++sub) {
            const float ta = tw + (float)(lane + 64 * sub) * dt2;
            if (ta < ra.tMax) {
                // g = G + E t is affine in the pixel, so over the patch each component of E lies between its values on the
                // four corner rays, and over the slab (t >= 0) g_a lies between G_a + t eMin_a and G_a + t eMax_a at the
                // slab's ends: the exact hull of the beam's part, grown by the margin
                const float tA = __builtin_fmaxf(ta - 1.0e-4f * dt2, 0.0f), tB = ta + 1.0001f * dt2;
                int k0[3], k1[3];
                bool huge = false;
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const float gl = ra.G[a] + __builtin_fminf(tA * eMin[a], tB * eMin[a]);
                    const float gh = ra.G[a] + __builtin_fmaxf(tA * eMax[a], tB * eMax[a]);
                    const float m = 0.02f + 1.0e-5f * __builtin_fmaxf(__builtin_fabsf(gl), __builtin_fabsf(gh));
                    k0[a] = f2i_rz(__builtin_floorf(gl - m)) >> 3;
                    k1[a] = f2i_rz(__builtin_floorf(gh + m)) >> 3;
                    huge |= !(k1[a] - k0[a] <= 1) || !(gl == gl) || !(gh == gh);
                }
                const int r0 = k0[0] - base0, r1 = k0[1] - base1, r2 = k0[2] - base2;
                huge |= (uint32_t)r0 >= 1022u || (uint32_t)r1 >= 1022u || (uint32_t)r2 >= 1022u;
                if (huge) {
                    fail = true;
                } else {
                    uint32_t word[8], bit[8];
                    // (the eight hashes share their products: the second cell of an axis is the first plus one)
                    const uint32_t hx0 = (uint32_t)k0[0] * 73856093u, hy0 = (uint32_t)k0[1] * 19349669u, hz0 = (uint32_t)k0[2] * 83492791u;
                    const uint32_t hx1 = hx0 + (k1[0] != k0[0] ? 73856093u : 0u), hy1 = hy0 + (k1[1] != k0[1] ? 19349669u : 0u),
                                   hz1 = hz0 + (k1[2] != k0[2] ? 83492791u : 0u);
                    const bool pow2 = (fp.numBuckets & (fp.numBuckets - 1u)) == 0u;
                    uint32_t mineMask = 0u;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const uint32_t hh = ((k & 1) ? hx1 : hx0) ^ ((k & 2) ? hy1 : hy0) ^ ((k & 4) ? hz1 : hz0);      // calculateHash, VoxelUtils.cu:250-259
                        const uint32_t h = pow2 ? hh & (fp.numBuckets - 1u) : hh % fp.numBuckets;
                        // (a bucket of another shard reads word 0 and masks the bit out: no branch around the load)
                        const bool mine = h >= fp.bucketLo && h < fp.bucketHi;
                        const uint32_t local = mine ? h - fp.bucketLo : 0u;
                        word[k] = dp.bucketBits[local >> 5];
                        bit[k] = local & 31u;
                        mineMask |= mine ? 1u << k : 0u;
                    }
                    // (a box one block wide on an axis names each cell twice: only its first name is taken)
                    uint32_t setMask = 0u;
#pragma unroll
                    for (int k = 0; k < 8; ++k) setMask |= ((word[k] >> bit[k]) & 1u) << k;
                    setMask &= mineMask;
                    setMask &= ~((k1[0] == k0[0] ? 0xaau : 0u) | (k1[1] == k0[1] ? 0xccu : 0u) | (k1[2] == k0[2] ? 0xf0u : 0u));
                    while (setMask) {
                        const int k = __builtin_ctz(setMask);
                        setMask &= setMask - 1u;
                        const uint32_t tag = coop_tag(r0 + (k & 1), r1 + ((k >> 1) & 1), r2 + ((k >> 2) & 1));
                        uint32_t h = (tag * 2654435761u) >> 24;
                        bool placed = false;
                        for (int probe = 0; probe < 16 && !placed; ++probe) {
                            const uint32_t old = atomicCAS(&tags[h], 0u, tag);
                            if (old == 0u) {                       // a new cell: queued for step 2
                                ptrs[h] = kCoopUnresolved;
                                cells[atomicAdd(count, 1u)] = (uint16_t)h;
                            }
                            placed = old == 0u || old == tag;
                            h = (h + 1u) & (kCoopSlots - 1);
                        }
                        if (!placed) fail = true;
                    }
                }
            }
            }
            fail = __ballot(fail) != 0ull;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (fail) break;
            if (ra.stamps) stampA = __builtin_amdgcn_s_memrealtime();
            // ---- 2. the new cells: allocated? ----
            const int listBegin = nList;
            const int cellBegin = nCells;
            nCells = __builtin_amdgcn_readfirstlane((int)*count);
            // Wave-cooperative bucket scan: eight lanes per cell, lane j of a group reads slot j (+8, ...) of the cell's bucket,
            // so a bucket costs one round trip however full it is (getVoxelEntry4Block's slot loop, VoxelUtils.cu:374-381,
            // turned sideways; keys are unique, so at most one lane of a group matches and it publishes the pointer; the
            // group's share of the ballot says whether anyone did).  The chain behind the bucket's last slot (:384-411, overflow
            // list) is a linked list: one lane of the group follows it.
            for (int cb = cellBegin; cb < nCells; cb += 8) {
                const int ci = cb + (lane >> 3);
                const uint32_t sub = (uint32_t)lane & 7u;
                const bool has = ci < nCells;
                const int slot = has ? (int)cells[ci] : 0;
                const uint32_t t = tags[slot] - 1u;
                const int qx = base0 + (int)(t & 1023u), qy = base1 + (int)((t >> 10) & 1023u), qz = base2 + (int)(t >> 20);
                const uint32_t myLocal = hash_block(qx, qy, qz, fp.numBuckets) - fp.bucketLo;      // (a set bit: the bucket is this shard's)
                const uint32_t start = myLocal * fp.bucketSize;
                bool foundHere = false;
                for (uint32_t sb = 0; sb < fp.bucketSize; sb += 8u) {
                    const uint32_t i = sb + sub;
                    bool match = false;
                    if (has && i < fp.bucketSize) {
                        const VoxelEntry e = dp.table[start + i];
                        match = entry_is(e, qx, qy, qz);
                        if (match) ptrs[slot] = (uint32_t)e.ptr;
                    }
                    foundHere |= ((uint32_t)(__ballot(match) >> (lane & ~7)) & 0xffu) != 0u;
                }
                if (has && !foundHere && sub == 0u) {
                    int ptr = VH_FREE_BLOCK;
                    if (fp.flags & kFlagOverflow) {
                        const uint32_t last = start + fp.bucketSize - 1u, n = owned_entries(fp);
                        uint32_t i = last;
                        for (uint32_t iter = 0; iter < fp.listSize; ++iter) {                 // :391-392
                            const VoxelEntry curr = dp.table[i];
                            if (entry_is(curr, qx, qy, qz)) { ptr = curr.ptr; break; }
                            if (curr.offset == 0) break;                                      // :396
                            i = chain_slot(last, curr.offset, n);                             // :398-399
                        }
                    }
                    ptrs[slot] = (uint32_t)ptr;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int cb = cellBegin; cb < nCells; cb += 64) {
                const bool todo = cb + lane < nCells;
                const int slot = todo ? (int)cells[cb + lane] : 0;
                const uint32_t t = tags[slot] - 1u;
                const int qx = base0 + (int)(t & 1023u), qy = base1 + (int)((t >> 10) & 1023u), qz = base2 + (int)(t >> 20);
                const int ptr = todo ? (int)ptrs[slot] : VH_FREE_BLOCK;
                // the allocated ones join the wave's list
                const bool isNew = todo && ptr != VH_FREE_BLOCK;
                const unsigned long long m = __ballot(isNew);
                if (isNew) {
                    // camera depth of the block's centre in voxels beyond t_min: the order the blocks are walked in
                    const float zc = ((ra.zrow[0] * ((float)(qx << 3) + 3.5f) + ra.zrow[1] * ((float)(qy << 3) + 3.5f))
                                      + ra.zrow[2] * ((float)(qz << 3) + 3.5f)) + ra.zrow[3];
                    const float kq = __builtin_fminf(__builtin_fmaxf((zc - ra.tMin) * ra.invVs, 0.0f), 65535.0f);
                    list[nList + __popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)slot | ((uint32_t)f2i_rz(kq) << 16);
                }
                nList += __popcll(m);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // Front to back: a ray that has found its hit skips every block it enters after it, so the blocks behind the
            // surface are walked only by the rays that missed it (the outcome does not depend on the order: the earliest
            // arrival wins whichever block is judged first).  Rank = number of smaller words, all distinct.
            {
                const int n = nList - listBegin;
                if (n > 1 && n <= 64) {
                    const uint32_t mine = lane < n ? list[listBegin + lane] : 0xffffffffu;
                    int rank = 0;
                    for (int i = 0; i < n; ++i) rank += (uint32_t)__builtin_amdgcn_readlane((int)mine, i) < mine ? 1 : 0;
                    __builtin_amdgcn_wave_barrier();
                    if (lane < n) list[listBegin + rank] = mine;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
            }
            if (ra.stamps) { stampB = __builtin_amdgcn_s_memrealtime(); coopList = nList; }
            // The launch is as long as its slowest wave, and the slowest waves are the ones with the longest lists: they get the
            // issue slots first (s_setprio; 35.6 -> 32.6 us; thresholds 6/4/3, 7/5/3 and 10/7/5 measured the same).
            {
                const int n = nList - listBegin;
                if (n >= 8) __builtin_amdgcn_s_setprio(3); else if (n >= 6) __builtin_amdgcn_s_setprio(2); else if (n >= 4) __builtin_amdgcn_s_setprio(1);
            }
            // ---- 3. every ray against every new block of the list ----
            // (a wave-uniform loop: the block's key and voxel pointer are scalars.  Measured alternative: every ray walking
            // its OWN blocks, one per round -- the busiest ray of a wave enters as many blocks as the wave walks, 2.1 vs 2.2
            // rounds, and the per-lane block pointer made it 44.7 us against 40.2)
            {
                for (int i = listBegin; i < nList; ++i) {
                    if (__ballot(!final_) == 0ull) break;
                    // (measured in round 4: {tag, pointer} kept in walking order beside the list, one LDS round trip here instead of three
                    // dependent ones: 31.5 against 31.0 us -- 5 registers spilled instead of 2)
                    const int slot = __builtin_amdgcn_readfirstlane((int)(list[i] & 0xffffu));
                    const uint32_t tg = (uint32_t)__builtin_amdgcn_readfirstlane((int)tags[slot]) - 1u;
                    const int bptr = __builtin_amdgcn_readfirstlane((int)ptrs[slot]);
                    const int kk[3] = {base0 + (int)(tg & 1023u), base1 + (int)((tg >> 10) & 1023u), base2 + (int)(tg >> 20)};
                    const CoopEntry e = coop_entry(ax, c, kk, ra.tMax);
                    const float tE = e.tE;
                    const int pE = e.pE, xe = e.xe;
                    const bool inside = e.inside;
                    const bool enters = e.enters && !final_ && dda_before(tE, pE, bestT, bestP);      // (not behind the best candidate so far)
                    if (__ballot(enters) == 0ull) continue;
                    ++coopWalks;
                    if (!enters) continue;
                    // the voxel the ray enters at
                    int q[3];
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        const int lo = kk[a] << 3, hi = lo + 7;
                        const int nearC = ax[a].s > 0 ? lo : hi, farC = ax[a].s > 0 ? hi : lo;
                        if (inside || ax[a].invE == 0.0f) q[a] = c[a];
                        else if (a == xe) q[a] = nearC;
                        else q[a] = dda_advance(ax[a], prio[a], e.startIn[a] ? c[a] : nearC, farC, tE, pE);
                    }
                    // Inside the block the walk keeps the voxel as float coordinates (exact below 2^24: the host refuses views
                    // beyond 2^23), the local position as three 5-bit fields (value 8..15 = inside: ONE mask test tells when the
                    // ray has left the block) packed with the linear voxel index, and steps all of it without a branch:
                    // ~27 vector instructions per voxel instead of ~45 with integer coordinates and per-axis branches.
                    const int bx0 = kk[0] << 3, by0 = kk[1] << 3, bz0 = kk[2] << 3;
                    int pl;
                    {
                        const int lx = q[0] & 7, ly = q[1] & 7, lz = q[2] & 7;
                        pl = ((lx | (ly << 3) | (lz << 6)) << 16) | (lx + 8) | ((ly + 8) << 5) | ((lz + 8) << 10);
                    }
                    float fc0 = (float)q[0], fc1 = (float)q[1], fc2 = (float)q[2];
                    float tn0 = (fc0 - gs0) * ie0, tn1 = (fc1 - gs1) * ie1, tn2 = (fc2 - gs2) * ie2;
                    float tArr = tE;
                    int pArr = pE;
                    bool pv = false, firstVoxel = !inside, walking = true;
                    float ps = 0.0f;
                    int prevLin = -1;                             // the previous sample: a voxel of this block, or (-1) the neighbour behind the entry face
                    const Voxel *blk = dp.blocks + (size_t)bptr;
                    while (walking) {
                        int pls[kCoopK], vp[kCoopK];
                        float vt[kCoopK];
                        Voxel vv[kCoopK];
                        int n = 0;
                        bool more = true;
#pragma unroll
                        for (int j = 0; j < kCoopK; ++j) {
                            if (more) {
                                pls[j] = pl; vt[j] = tArr; vp[j] = pArr;
                                vv[j] = blk[(uint32_t)pl >> 16];
                                n = j + 1;
                                // the crossing that ends this voxel (raycastSDF.frag:156-170)
                                const bool m0 = tn0 < tn1 && tn0 < tn2;
                                const bool m2 = !m0 && tn2 < tn1;
                                const bool m1 = !m0 && !m2;
                                tArr = m0 ? tn0 : m2 ? tn2 : tn1;
                                pArr = m0 ? 2 : m2 ? 1 : 0;
                                pl += m0 ? d0 : m2 ? d2 : d1;
                                fc0 += m0 ? f0 : 0.0f; fc1 += m1 ? f1 : 0.0f; fc2 += m2 ? f2 : 0.0f;
                                tn0 = (fc0 - gs0) * ie0; tn1 = (fc1 - gs1) * ie1; tn2 = (fc2 - gs2) * ie2;
                                more = tArr < ra.tMax && (pl & 0x6318) == 0x2108;     // (a voxel is visited iff the ray arrives before t_max)
                            }
                        }
                        walking = more;
#pragma unroll
                        for (int j = 0; j < kCoopK; ++j) {
                            if (j < n) {
                                const bool valid = vv[j].weight > 0.0f;
                                const int lin = (int)((uint32_t)pls[j] >> 16);
                                if (valid && vv[j].sdf <= 0.0f) {
                                    if (firstVoxel) {
                                        const int vx = bx0 + (lin & 7), vy = by0 + ((lin >> 3) & 7), vz = bz0 + (lin >> 6);
                                        // the voxel the ray was in before the entry event: one step back on the entry axis, in the
                                        // neighbouring block -- allocated iff it is in the wave's set
                                        const int n0 = vx - (xe == 0 ? ax[0].s : 0), n1 = vy - (xe == 1 ? ax[1].s : 0), n2 = vz - (xe == 2 ? ax[2].s : 0);
                                        const int fs = coop_find(tags, coop_tag((n0 >> 3) - base0, (n1 >> 3) - base1, (n2 >> 3) - base2));
                                        pv = false;
                                        if (fs >= 0) {
                                            const uint32_t np = ptrs[fs];
                                            if (np != (uint32_t)VH_FREE_BLOCK && np != kCoopUnresolved) {
                                                const Voxel nb = dp.blocks[(size_t)np + (size_t)(((n2 & 7) << 6) | ((n1 & 7) << 3) | (n0 & 7))];
                                                pv = nb.weight > 0.0f; ps = nb.sdf; prevLin = -1;
                                            }
                                        }
                                    }
                                    if (pv && ps > 0.0f) {
                                        if (dda_before(vt[j], vp[j], bestT, bestP)) {
                                            // Only WHERE the pair sits is kept here (the set's slot, the voxel, the previous sample, the entry
                                            // axis) with its two values; the depth is worked out once, behind the last block (below).  On a
                                            // grazing patch some ray finds its pair at nearly every step of the wave, and the dot products and
                                            // the division under this branch then doubled the step's instructions.
                                            bestT = vt[j]; bestP = vp[j];
                                            recPs = ps; recSdf = vv[j].sdf;
                                            recW = slot | (lin << 8) | ((prevLin & 1023) << 17) | (xe << 27);
                                        }
                                        walking = false;                   // (the block's first pair: nothing earlier behind it)
                                        n = j;                             // (stops the judging)
                                    }
                                }
                                pv = valid; ps = vv[j].sdf; prevLin = lin;
                                firstVoxel = false;
                            }
                        }
                    }
                }
            }
            // a candidate that arrived before this window's end cannot be beaten by a block found later
            final_ = final_ || bestT < tw + window;
            if (__ballot(!final_) == 0ull) break;
        }
        if (!fail) {
            coopDone = true; live = false;
            if (recW != -1) {
                // the best candidate: its voxel and the previous sample's, from the set's slot
                const int slot = recW & 255, lin = (recW >> 8) & 511, pl = (recW >> 17) & 1023, xe = (recW >> 27) & 3;
                const uint32_t tg = tags[slot] - 1u;
                const int b0 = (base0 + (int)(tg & 1023u)) << 3, b1 = (base1 + (int)((tg >> 10) & 1023u)) << 3, b2 = (base2 + (int)(tg >> 20)) << 3;
                const int vx = b0 + (lin & 7), vy = b1 + ((lin >> 3) & 7), vz = b2 + (lin >> 6);
                const bool nb = pl == 1023;            // the voxel the ray was in before the block's entry event
                const int p0 = nb ? vx - (xe == 0 ? ax[0].s : 0) : b0 + (pl & 7);
                const int p1 = nb ? vy - (xe == 1 ? ax[1].s : 0) : b1 + ((pl >> 3) & 7);
                const int p2 = nb ? vz - (xe == 2 ? ax[2].s : 0) : b2 + (pl >> 6);
                // the samples sit at their voxels' centres: camera depth = row 2 of the inverse pose
                const float tc = ((ra.zrow[0] * (float)vx + ra.zrow[1] * (float)vy) + ra.zrow[2] * (float)vz) + ra.zrow[3];
                const float tp = ((ra.zrow[0] * (float)p0 + ra.zrow[1] * (float)p1) + ra.zrow[2] * (float)p2) + ra.zrow[3];
                hit = tp + ((tc - tp) * recPs) / (recPs - recSdf);
                found = true;
                hx = vx; hy = vy; hz = vz; hptr = (int)ptrs[slot];
            }
        }
    }    // ---- a wave whose preconditions failed: the per-lane walk behind the beam front end ----
    unsigned long long stamp1 = stamp0;
    int budgetUsed = 0, round = 0;
    if (!coopDone) {
        dda_front_end(fp, dp, ra, dx, dy, ax, c, live);
        stamp1 = ra.stamps ? __builtin_amdgcn_s_memrealtime() : 0ull;
        const DdaHit h = dda_lane_walk<2>(fp, dp, ra, ax, c, live);
        hit = h.hit; found = h.found; hx = h.hx; hy = h.hy; hz = h.hz; hptr = h.hptr;
        budgetUsed = h.steps; round = h.rounds;
    }
    if (ra.stamps && lane == 0) {
        const size_t w = ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * kDdaBlockWaves + wave) * 8;
        ra.stamps[w] = stamp0; ra.stamps[w + 1] = __builtin_amdgcn_s_memrealtime();
        ra.stamps[w + 2] = coopDone ? (stampP1 - stamp0) : (unsigned long long)budgetUsed | ((unsigned long long)round << 32); ra.stamps[w + 3] = (unsigned long long)(pu | (pv << 16)) | ((stamp1 - stamp0) << 32);
        ra.stamps[w + 4] = stampA - stamp0; ra.stamps[w + 5] = stampB - stamp0; ra.stamps[w + 6] = (unsigned long long)coopList; ra.stamps[w + 7] = (unsigned long long)coopWalks;
    }
    if (!inImage) return;
    depthOut[(size_t)v * fp.width + u] = hit;
    if (!kNormals) return;
    // ---- normal of the hit: TSDF gradient at the hit voxel, normalised, camera frame, w = 0 ----
    float4 n = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (found) n = dda_normal(fp, dp, hx, hy, hz, hptr);
    normalOut[(size_t)v * fp.width + u] = n;
}

}  // namespace vh
