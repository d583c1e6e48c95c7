// vh_raycast_coop.hip -- the cooperative DDA raycast: one block list per 8x8 ray patch, the lists of a workgroup's four
// patches walked by whichever of its waves is free.  Part of libvoxelhash_hip.so (gfx950); included after vh_raycast.hip.
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
//
// Round 6: the listed blocks are ITEMS that an idle wave of the workgroup can take.  Measured before
// (profiles/r05_raycast_stamps.txt): a wave that walks its own list is as long as that list -- 3.5 us per block, lists of 8-13 blocks
// on silhouette and grazing patches against a mean of 2.7-4.2 -- and the launch is as long as its slowest wave (30-40 us, mean
// wave 15-19).  Sharing among the four neighbouring patches of a 16x16 tile gains next to nothing, because a grazing region covers
// all four.  So
//   (i) a workgroup's four patches are taken far apart: wave j of group g renders a patch of the j-th quarter of the row-major
//       patch grid, rotated by j quarter-rows inside its quarter (a quarter of the image away in both directions: a grazing wall
//       is a vertical band as often as a grazing floor is a horizontal one), so the patches of one grazing region meet light company;
//  (ii) every wave lists its own patch (steps 1-2), leaves its rays' set-up in LDS (E, 1/E, start voxel: 9 words per ray) and
//       publishes the list; it then walks its own list from the front with the loop of the one-list-per-wave form -- the rays'
//       state and best candidates in registers -- except that it TAKES every item from the patch's counter in LDS;
// (iii) a wave whose own list is done takes items of the neighbour with the most items left (same counter), walks the item's
//       block for the 64 rays of the item's patch from the state in LDS (coop_walk_item) and merges each ray's candidate into the
//       ray's 64-bit word in LDS with an atomic minimum: a candidate is {arrival event of the hit voxel (t, axis priority), where
//       the pair sits}, a ray's events are totally ordered, so the minimum IS the hit the sequential walk finds first, whatever the
//       order the blocks are walked in and whoever walks them; it runs at the priority of the list it helps with;
//  (iv) a patch nobody took from is written straight from its owner's registers; otherwise the owner merges its own best
//       candidates into the words once, and whoever completes the patch's last item turns the words into depth (and normals).
//       No barrier anywhere behind the first one; a wave with nothing left to take leaves.
// Same box, C2, mean over the bench's 50 poses: 28.8 us against 31.7 for one list per wave (33.9 / 35.6 with normals; 5 mm voxels,
// cooperative form forced: 163 / 172).  What did NOT work on the way (profiles/r06_raycast_*.txt, DESIGN_LOG.md round 6): every
// item through LDS, own ones too (33.2 us: 4.1 instead of 3.5 us per item); the owner publishing its best candidate after every
// block (30.1 against 28.6); helping only lists with 2 / 3 / 4 / 6 items left (29.2 ... 34.4); the four patches of one tile
// (31.6); forgetting the priorities by list length (+3 us).
constexpr int kCoopK = 1;                             // voxels fetched per round trip in the owner's block walk (2, 3: slower, rounds 4 and 6)
constexpr int kCoopSubs = 4;                          // slabs per lane and window of step 1 (64 x this many half-block slabs)
constexpr int kCoopSlots = 256;                       // per wave: cells with a set bucket bit
constexpr uint32_t kCoopUnresolved = 0x7ffffffeu;     // sPtr: not looked up yet
constexpr unsigned long long kCoopNone = ~0ull;       // a ray's word: no candidate
struct CoopShared {                                   // per workgroup (25 KiB: five workgroups per CU at 5 waves per SIMD)
    uint32_t tag[kDdaBlockWaves][kCoopSlots];      // per patch: the set of cells with a set bucket bit (the tag IS the key)
    uint32_t ptr[kDdaBlockWaves][kCoopSlots];      // ... and the voxel pointer of each (VH_FREE_BLOCK: not allocated)
    uint32_t list[kDdaBlockWaves][kCoopSlots];     // allocated cells: slot | depth key << 16, front to back -- the patch's ITEMS
    uint16_t cells[kDdaBlockWaves][kCoopSlots];    // the set's occupied slots in order of insertion (what step 2 resolves)
    float state[kDdaBlockWaves][9][64];            // the patch's rays: E[3], invE[3] (0: the axis never steps), c[3] (int bits)
    unsigned long long best[kDdaBlockWaves][64];   // the rays' best candidates {ordered t : 32 | priority : 2 | slot : 8 | voxel : 9 | previous : 10 | axis : 2 | 0}
    uint32_t count[kDdaBlockWaves];
    int base[kDdaBlockWaves][3];                   // block coordinates the tags are relative to
    int origin[kDdaBlockWaves];                    // pixel of the patch's first ray: u | v << 16
    uint32_t avail[kDdaBlockWaves];                // items of the patch (published with `ready`)
    uint32_t taken[kDdaBlockWaves];                // ... handed out so far (may run past avail)
    uint32_t done[kDdaBlockWaves];                 // ... walked to the end: the wave that completes a patch writes its pixels
    uint32_t ready[kDdaBlockWaves];                // the patch's list is final
};

// candidate word of a ray: events compare as (t, priority); t as a sign-ordered 32-bit key (-0 is written as +0)
__device__ __forceinline__ uint32_t rc_time_key(float t)
{
    const uint32_t b = __float_as_uint(t + 0.0f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float rc_key_time(uint32_t k)
{
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

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

// Step 3 for ONE item: the 64 rays of patch `w` against block `slot` of the patch's set.
// kK: voxels fetched per round trip of the walk (the path through the block is arithmetic alone, so the next kK voxels are
// enumerated first, their loads issued together, then judged in order).
template <int kK>
__device__ __forceinline__ void coop_walk_item(const FrameParams &fp, const DevPtrs &dp, const RaycastArgs &ra, CoopShared &sh, int w, int k,
                                               bool &entered)
{
    const int lane = threadIdx.x & 63;
    const int prio[3] = {2, 0, 1};
    const uint32_t *tags = sh.tag[w], *ptrs = sh.ptr[w];
    const int slot = __builtin_amdgcn_readfirstlane((int)(sh.list[w][k] & 0xffffu));
    const uint32_t tg = (uint32_t)__builtin_amdgcn_readfirstlane((int)tags[slot]) - 1u;
    const int bptr = __builtin_amdgcn_readfirstlane((int)ptrs[slot]);
    const int base0 = __builtin_amdgcn_readfirstlane(sh.base[w][0]), base1 = __builtin_amdgcn_readfirstlane(sh.base[w][1]),
              base2 = __builtin_amdgcn_readfirstlane(sh.base[w][2]);
    const int kk[3] = {base0 + (int)(tg & 1023u), base1 + (int)((tg >> 10) & 1023u), base2 + (int)(tg >> 20)};
    const int org = __builtin_amdgcn_readfirstlane(sh.origin[w]);
    const bool inImage = (org & 0xffff) + (lane & 7) < fp.width && (org >> 16) + (lane >> 3) < fp.height;
    // the ray's set-up, as its patch's wave left it
    DdaAxis ax[3];
    int c[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        ax[a].G = ra.G[a];
        ax[a].E = sh.state[w][a][lane];
        ax[a].invE = sh.state[w][3 + a][lane];
        c[a] = __float_as_int(sh.state[w][6 + a][lane]);
        ax[a].s = ax[a].E > 0.0f ? 1 : -1;
        ax[a].Gs = ax[a].E > 0.0f ? ax[a].G - 1.0f : ax[a].G;
    }
    unsigned long long *word = &sh.best[w][lane];
    const unsigned long long cur = *word;
    const float bestT = cur == kCoopNone ? __builtin_inff() : rc_key_time((uint32_t)(cur >> 32));
    const int bestP = cur == kCoopNone ? 3 : (int)((cur >> 30) & 3ull);
    const CoopEntry e = coop_entry(ax, c, kk, ra.tMax);
    const float tE = e.tE;
    const int pE = e.pE, xe = e.xe;
    const bool inside = e.inside;
    const bool enters = inImage && e.enters && dda_before(tE, pE, bestT, bestP);      // (not behind the candidate the ray holds)
    entered = __ballot(enters) != 0ull;
    if (!enters) return;
    // the walk's per-ray constants
    const float f0 = (float)ax[0].s, f1 = (float)ax[1].s, f2 = (float)ax[2].s;
    // (an axis that never steps: invE = 0 would give a crossing time of 0; (c + 1e30) * inf = inf instead)
    const float ie0 = ax[0].invE != 0.0f ? ax[0].invE : __builtin_inff(), gs0 = ax[0].invE != 0.0f ? ax[0].Gs : -1.0e30f;
    const float ie1 = ax[1].invE != 0.0f ? ax[1].invE : __builtin_inff(), gs1 = ax[1].invE != 0.0f ? ax[1].Gs : -1.0e30f;
    const float ie2 = ax[2].invE != 0.0f ? ax[2].invE : __builtin_inff(), gs2 = ax[2].invE != 0.0f ? ax[2].Gs : -1.0e30f;
    const int d0 = ax[0].s * ((1 << 16) + 1), d1 = ax[1].s * ((8 << 16) + (1 << 5)), d2 = ax[2].s * ((64 << 16) + (1 << 10));
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
    bool pvd = false, firstVoxel = !inside, walking = true, have = false;
    float ps = 0.0f, candT = 0.0f;
    int prevLin = -1, candP = 0, candLin = 0, candPrev = 0;      // prevLin: a voxel of this block, or (-1) the neighbour behind the entry face
    const Voxel *blk = dp.blocks + (size_t)bptr;
    while (walking) {
        int pls[kK], vp[kK];
        float vt[kK];
        Voxel vv[kK];
        int n = 0;
        bool more = true;
#pragma unroll
        for (int j = 0; j < kK; ++j) {
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
        for (int j = 0; j < kK; ++j) {
            if (j < n) {
                const bool valid = vv[j].weight > 0.0f;
                const int lin = (int)((uint32_t)pls[j] >> 16);
                if (valid && vv[j].sdf <= 0.0f) {
                    if (firstVoxel) {
                        // the voxel the ray was in before the entry event: one step back on the entry axis, in the neighbouring
                        // block -- allocated iff it is in the patch's set
                        const int vx = bx0 + (lin & 7), vy = by0 + ((lin >> 3) & 7), vz = bz0 + (lin >> 6);
                        const int n0 = vx - (xe == 0 ? ax[0].s : 0), n1 = vy - (xe == 1 ? ax[1].s : 0), n2 = vz - (xe == 2 ? ax[2].s : 0);
                        const int fs = coop_find(tags, coop_tag((n0 >> 3) - base0, (n1 >> 3) - base1, (n2 >> 3) - base2));
                        pvd = false;
                        if (fs >= 0) {
                            const uint32_t np = ptrs[fs];
                            if (np != (uint32_t)VH_FREE_BLOCK && np != kCoopUnresolved) {
                                const Voxel nb = dp.blocks[(size_t)np + (size_t)(((n2 & 7) << 6) | ((n1 & 7) << 3) | (n0 & 7))];
                                pvd = nb.weight > 0.0f; ps = nb.sdf; prevLin = -1;
                            }
                        }
                    }
                    if (pvd && ps > 0.0f) {                    // the block's first pair: nothing earlier behind it
                        have = dda_before(vt[j], vp[j], bestT, bestP);
                        candT = vt[j]; candP = vp[j]; candLin = lin; candPrev = prevLin & 1023;
                        walking = false;
                        n = j;                                 // (stops the judging)
                    }
                }
                pvd = valid; ps = vv[j].sdf; prevLin = lin;
                firstVoxel = false;
            }
        }
    }
    if (have) {
        // Only WHERE the pair sits travels in the word (the set's slot, the voxel, the previous sample, the entry axis); depth
        // and normal are worked out once per ray, by the wave that completes the patch.
        const unsigned long long cand = ((unsigned long long)rc_time_key(candT) << 32) | ((unsigned long long)candP << 30) |
                                        ((unsigned long long)slot << 22) | ((unsigned long long)candLin << 13) |
                                        ((unsigned long long)candPrev << 3) | ((unsigned long long)xe << 1);
        (void)__hip_atomic_fetch_min(word, cand, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// A completed patch: its rays' words become depth (and normals), with the sequential walk's arithmetic.
template <bool kNormals>
__device__ __forceinline__ void coop_resolve_patch(const FrameParams &fp, const DevPtrs &dp, const RaycastArgs &ra, CoopShared &sh, int w,
                                                   float *__restrict__ depthOut, float4 *__restrict__ normalOut)
{
    const int lane = threadIdx.x & 63;
    const int org = __builtin_amdgcn_readfirstlane(sh.origin[w]);
    const int u = (org & 0xffff) + (lane & 7), v = (org >> 16) + (lane >> 3);
    if (u >= fp.width || v >= fp.height) return;
    const unsigned long long word = sh.best[w][lane];
    float hit = 0.0f;
    float4 n = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (word != kCoopNone) {
        const uint32_t *tags = sh.tag[w], *ptrs = sh.ptr[w];
        const int slot = (int)((word >> 22) & 255ull), lin = (int)((word >> 13) & 511ull), pl = (int)((word >> 3) & 1023ull),
                  xe = (int)((word >> 1) & 3ull);
        const uint32_t tg = tags[slot] - 1u;
        const int base0 = sh.base[w][0], base1 = sh.base[w][1], base2 = sh.base[w][2];
        const int b0 = (base0 + (int)(tg & 1023u)) << 3, b1 = (base1 + (int)((tg >> 10) & 1023u)) << 3, b2 = (base2 + (int)(tg >> 20)) << 3;
        const int vx = b0 + (lin & 7), vy = b1 + ((lin >> 3) & 7), vz = b2 + (lin >> 6);
        const bool nb = pl == 1023;            // the voxel the ray was in before the block's entry event
        const int s = sh.state[w][xe][lane] > 0.0f ? 1 : -1;      // the entry axis' direction (E > 0)
        const int p0 = nb ? vx - (xe == 0 ? s : 0) : b0 + (pl & 7);
        const int p1 = nb ? vy - (xe == 1 ? s : 0) : b1 + ((pl >> 3) & 7);
        const int p2 = nb ? vz - (xe == 2 ? s : 0) : b2 + (pl >> 6);
        const int hptr = (int)ptrs[slot];
        int pptr = hptr;
        if (nb) pptr = (int)ptrs[coop_find(tags, coop_tag((p0 >> 3) - base0, (p1 >> 3) - base1, (p2 >> 3) - base2))];     // (in the set: the walk found it there)
        const float recSdf = dp.blocks[(size_t)hptr + (size_t)lin].sdf;
        const float recPs = dp.blocks[(size_t)pptr + (size_t)(((p2 & 7) << 6) | ((p1 & 7) << 3) | (p0 & 7))].sdf;
        // the samples sit at their voxels' centres: camera depth = row 2 of the inverse pose
        const float tc = ((ra.zrow[0] * (float)vx + ra.zrow[1] * (float)vy) + ra.zrow[2] * (float)vz) + ra.zrow[3];
        const float tp = ((ra.zrow[0] * (float)p0 + ra.zrow[1] * (float)p1) + ra.zrow[2] * (float)p2) + ra.zrow[3];
        hit = tp + ((tc - tp) * recPs) / (recPs - recSdf);
        if (kNormals) n = dda_normal(fp, dp, vx, vy, vz, hptr);
    }
    depthOut[(size_t)v * fp.width + u] = hit;
    if (kNormals) normalOut[(size_t)v * fp.width + u] = n;
}

template <bool kNormals>
__global__ __launch_bounds__(64 * kDdaBlockWaves, VH_DDA_WAVES) void raycast_coop_kernel(const FrameParams fp, const DevPtrs dp, const RaycastArgs ra,
                                                           float *__restrict__ depthOut, float4 *__restrict__ normalOut)
{
    __shared__ CoopShared sh;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long stamp0 = ra.stamps ? __builtin_amdgcn_s_memrealtime() : 0ull;
    unsigned long long stampA = stamp0, stampB = stamp0;      // diagnostics: the set is built / the list is resolved
    // group -> patches: each XCD (workgroup index mod 8, own L2) takes a contiguous run of groups, a group's patches lie
    // `groups` apart in the row-major patch grid
    int g = (int)blockIdx.x;
    if ((ra.groups & 7) == 0) g = (g & 7) * (ra.groups >> 3) + (g >> 3);
    // wave j takes a patch of the j-th quarter of the row-major patch grid (a quarter of the rows further down), rotated inside
    // its quarter by j quarter-rows (a quarter of a row further right): a grazing wall is a vertical band as often as a grazing
    // floor is a horizontal one.  A rotation inside the quarter: every patch is still rendered exactly once.
    const int gq = (g + wave * (ra.patchesX >> 2)) % ra.groups;
    const int patch = gq + wave * ra.groups;
    const bool hasPatch = patch < ra.numPatches;
    const int py = patch / ra.patchesX, px = patch - py * ra.patchesX;
    const int pu = px * 8, pv = py * 8;
    const int u = pu + (lane & 7), v = pv + (lane >> 3);
    const bool inImage = hasPatch && u < fp.width && v < fp.height;
    {
        uint32_t *tags = sh.tag[wave];
#pragma unroll
        for (int r = 0; r < kCoopSlots / 64; ++r) tags[lane + 64 * r] = 0u;
        sh.best[wave][lane] = kCoopNone;
        if (lane == 0) {
            sh.count[wave] = 0u; sh.avail[wave] = 0u; sh.taken[wave] = 0u; sh.done[wave] = 0u; sh.ready[wave] = 0u;
            sh.origin[wave] = pu | (pv << 16);
        }
    }
    __syncthreads();
    int nList = 0, ownCount = 0, coopWalks = 0;
    bool ownWalk = false;
    auto publish = [&](int n) {          // the list, the rays' state and the set are in LDS before `ready` is
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) {
            __hip_atomic_store(&sh.avail[wave], (uint32_t)n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(&sh.ready[wave], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    bool published = false;
    DdaAxis ax[3];
    int c[3];
    float dx = 0.0f, dy = 0.0f;
    if (hasPatch) {
        // ---- steps 1 and 2 for this wave's own patch ----
        dda_ray(fp, ra, u, v, ax, c, dx, dy);
        const float vs = fp.voxelSize;
        uint32_t *tags = sh.tag[wave], *ptrs = sh.ptr[wave];
        uint32_t *list = sh.list[wave];
        uint16_t *cells = sh.cells[wave];
        uint32_t *count = &sh.count[wave];
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
        bool fail = false;
        __builtin_amdgcn_wave_barrier();
        // A window = the slabs one pass of step 1 covers: 64 per sub-pass, up to kCoopSubs sub-passes when the range is longer
        // (finer voxels).  One window for the whole range beats several (640x480, 60 frames, per call: 2 windows of 64 slabs
        // 65.5 us, 3: 83.6, 4: 102.0 -- every window pays the front end again, and the per-lane walk behind a beam front end,
        // 58.9 / 74.3 / 92.5, was faster); what lies behind a ray's hit is then listed too, but skipped by its arrival event.
        // A range of several windows (more than 1 024 voxels deep) lists them all before anything is walked.
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
            if (ra.stamps && tw == ra.tMin) stampA = __builtin_amdgcn_s_memrealtime();
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
            if (ra.stamps && tw == ra.tMin) stampB = __builtin_amdgcn_s_memrealtime();
        }
        if (fail) {
            // the preconditions failed (a box wider than two blocks, the set full, a block too far from the first): this patch
            // takes the per-lane walk behind the beam front end (below, once the neighbours know there is nothing to share)
            ownWalk = true;
            nList = 0;
        } else if (nList == 0) {
            if (inImage) {                       // no allocated block along any ray of the patch
                depthOut[(size_t)v * fp.width + u] = 0.0f;
                if (kNormals) normalOut[(size_t)v * fp.width + u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
        } else {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                sh.state[wave][a][lane] = ax[a].E;
                sh.state[wave][3 + a][lane] = ax[a].invE;
                sh.state[wave][6 + a][lane] = __int_as_float(c[a]);
            }
            if (lane == 0) { sh.base[wave][0] = base0; sh.base[wave][1] = base1; sh.base[wave][2] = base2; }
            publish(nList);
            published = true;
            // ---- step 3, the owner's loop: this wave walks its own list from the front with its rays' state in registers (the
            // loop of the one-list-per-wave form); every item is TAKEN from the patch's counter, which idle neighbours take from too
            const int prio[3] = {2, 0, 1};
            const float f0 = (float)ax[0].s, f1 = (float)ax[1].s, f2 = (float)ax[2].s;
            // (an axis that never steps: invE = 0 would give a crossing time of 0; (c + 1e30) * inf = inf instead)
            const float ie0 = ax[0].invE != 0.0f ? ax[0].invE : __builtin_inff(), gs0 = ax[0].invE != 0.0f ? ax[0].Gs : -1.0e30f;
            const float ie1 = ax[1].invE != 0.0f ? ax[1].invE : __builtin_inff(), gs1 = ax[1].invE != 0.0f ? ax[1].Gs : -1.0e30f;
            const float ie2 = ax[2].invE != 0.0f ? ax[2].invE : __builtin_inff(), gs2 = ax[2].invE != 0.0f ? ax[2].Gs : -1.0e30f;
            const int d0 = ax[0].s * ((1 << 16) + 1), d1 = ax[1].s * ((8 << 16) + (1 << 5)), d2 = ax[2].s * ((64 << 16) + (1 << 10));
            float bestT = __builtin_inff();                            // arrival event of the best candidate's hit voxel
            int bestP = 3;
            int recW = -1;                                             // where the best candidate's pair sits (-1: none yet)
            float recPs = 0.0f, recSdf = 0.0f;                         // its two samples
            // The launch is as long as its slowest waves, and those are the ones with the longest lists: they get the issue slots first
            // (s_setprio; 35.6 -> 32.6 us in round 3).  A wave that helps with a long list runs at that list's priority (below).
            if (nList >= 8) __builtin_amdgcn_s_setprio(3); else if (nList >= 6) __builtin_amdgcn_s_setprio(2); else if (nList >= 4) __builtin_amdgcn_s_setprio(1);
            for (;;) {
                uint32_t tk = 0u;
                if (lane == 0) tk = __hip_atomic_fetch_add(&sh.taken[wave], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const int t = __builtin_amdgcn_readfirstlane((int)tk);
                if (t >= nList) break;
                ++ownCount;
                {
            const int slot = __builtin_amdgcn_readfirstlane((int)(list[t] & 0xffffu));
            const uint32_t tg = (uint32_t)__builtin_amdgcn_readfirstlane((int)tags[slot]) - 1u;
            const int bptr = __builtin_amdgcn_readfirstlane((int)ptrs[slot]);
            const int kk[3] = {base0 + (int)(tg & 1023u), base1 + (int)((tg >> 10) & 1023u), base2 + (int)(tg >> 20)};
            const CoopEntry e = coop_entry(ax, c, kk, ra.tMax);
            const float tE = e.tE;
            const int pE = e.pE, xe = e.xe;
            const bool inside = e.inside;
            const bool enters = inImage && e.enters && dda_before(tE, pE, bestT, bestP);      // (not behind the best candidate so far)
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
            if (ownCount == nList) {
                // nobody took an item of this patch: depth (and normals) straight from the registers, as the one-list-per-wave form
                float hit = 0.0f;
                bool found = false;
                int hx = 0, hy = 0, hz = 0, hptr = VH_FREE_BLOCK;
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
                if (inImage) {
                    depthOut[(size_t)v * fp.width + u] = hit;
                    if (kNormals) normalOut[(size_t)v * fp.width + u] = found ? dda_normal(fp, dp, hx, hy, hz, hptr) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                }
            } else {
                // neighbours walked some of the items: the rays' words hold everybody's candidates; whoever completes the patch writes it
                if (recW != -1) {         // (this wave's own best candidate joins them: once, not per block -- per block it cost 1.5 us)
                    const unsigned long long cand = ((unsigned long long)rc_time_key(bestT) << 32) | ((unsigned long long)bestP << 30) |
                                                    ((unsigned long long)(recW & 255) << 22) | ((unsigned long long)((recW >> 8) & 511) << 13) |
                                                    ((unsigned long long)((recW >> 17) & 1023) << 3) | ((unsigned long long)((recW >> 27) & 3) << 1);
                    (void)__hip_atomic_fetch_min(&sh.best[wave][lane], cand, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                uint32_t d = 0u;
                if (lane == 0) d = __hip_atomic_fetch_add(&sh.done[wave], (uint32_t)ownCount, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP) + (uint32_t)ownCount;
                d = (uint32_t)__builtin_amdgcn_readfirstlane((int)d);
                if (d == (uint32_t)nList) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    coop_resolve_patch<kNormals>(fp, dp, ra, sh, wave, depthOut, normalOut);
                }
            }
        }
    }
    if (!published) publish(0);          // (no patch, an empty list, or the per-lane fall-back: nothing to share)
    if (ownWalk) {
        bool live = inImage;
        dda_front_end(fp, dp, ra, dx, dy, ax, c, live);
        const DdaHit h = dda_lane_walk<1>(fp, dp, ra, ax, c, live);
        if (inImage) {
            depthOut[(size_t)v * fp.width + u] = h.hit;
            if (kNormals) normalOut[(size_t)v * fp.width + u] = h.found ? dda_normal(fp, dp, h.hx, h.hy, h.hz, h.hptr) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
    }
    // ---- step 3, the idle wave's loop: take items of the neighbours' patches, walk them through LDS, complete patches ----
    int nTaken = 0, nWalked = 0;
    for (;;) {
        int w = -1, k = 0;
        bool allReady = true;
        // the neighbour with the most items left (thresholds of 2, 3, 4, 6 items before a list is helped with: each slower)
        int most = 0;
        for (int r = 1; r < kDdaBlockWaves; ++r) {
            const int w2 = (wave + r) & (kDdaBlockWaves - 1);
            if (__hip_atomic_load(&sh.ready[w2], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u) { allReady = false; continue; }
            const int left = (int)__hip_atomic_load(&sh.avail[w2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) -
                             (int)__hip_atomic_load(&sh.taken[w2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (left > most) { most = left; w = w2; }
        }
        if (w >= 0) {
            const uint32_t av = __hip_atomic_load(&sh.avail[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            uint32_t t = 0u;
            if (lane == 0) t = __hip_atomic_fetch_add(&sh.taken[w], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
            if (t < av) k = (int)t;
            else continue;                        // (the owner or another neighbour was faster: look again)
        }
        if (w < 0) {
            if (allReady) break;
            __builtin_amdgcn_s_sleep(8);          // a neighbour is still listing its patch
            continue;
        }
        ++nTaken;
        if (most >= 6) __builtin_amdgcn_s_setprio(3); else if (most >= 4) __builtin_amdgcn_s_setprio(2); else if (most >= 2) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
        bool entered = false;
        coop_walk_item<1>(fp, dp, ra, sh, w, k, entered);
        nWalked += entered ? 1 : 0;
        // the item is done when its candidates are in the rays' words
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        uint32_t d = 0u;
        if (lane == 0) d = __hip_atomic_fetch_add(&sh.done[w], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP) + 1u;
        d = (uint32_t)__builtin_amdgcn_readfirstlane((int)d);
        if (d == __hip_atomic_load(&sh.avail[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            coop_resolve_patch<kNormals>(fp, dp, ra, sh, w, depthOut, normalOut);
        }
    }
    if (ra.stamps && lane == 0) {
        const size_t wd = ((size_t)blockIdx.x * kDdaBlockWaves + wave) * 8;
        ra.stamps[wd] = stamp0; ra.stamps[wd + 1] = __builtin_amdgcn_s_memrealtime();
        ra.stamps[wd + 2] = (unsigned long long)ownCount | ((unsigned long long)nTaken << 32); ra.stamps[wd + 3] = (unsigned long long)(pu | (pv << 16));
        ra.stamps[wd + 4] = stampA - stamp0; ra.stamps[wd + 5] = stampB - stamp0; ra.stamps[wd + 6] = (unsigned long long)nList;
        ra.stamps[wd + 7] = (unsigned long long)coopWalks | ((unsigned long long)nWalked << 32);
    }
}

}  // namespace vh
