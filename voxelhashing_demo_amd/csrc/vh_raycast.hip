// vh_raycast.hip -- raycast through the hash (stand-in for SDFRenderer::render).
// Part of libvoxelhash_hip.so (gfx950); included by vh_kernels.hip after vh_device.h.
#pragma once

namespace vh {

// ---------------------------------------------------------------------------
// raycast
// ---------------------------------------------------------------------------
// getVoxelEntry4Block, live half (VoxelUtils.cu:362-382)
__device__ __forceinline__ int lookup_block(const FrameParams &fp, const DevPtrs &dp, int kx, int ky, int kz)
{
    const uint32_t h = hash_block(kx, ky, kz, fp.numBuckets);
    if (h < fp.bucketLo || h >= fp.bucketHi) return VH_FREE_BLOCK;
    const uint32_t local = h - fp.bucketLo;
    // One bit per bucket ("holds at least one entry", set by the commit phase): a few hundred
    // KB that stay in L2, while the table itself is >100 MB.  Nearly every block a ray crosses
    // is empty space and is answered here without touching the table.
    if (!((dp.bucketBits[local >> 5] >> (local & 31u)) & 1u)) return VH_FREE_BLOCK;
    if (fp.flags & kFlagOverflow) {
        // (a bucket with a chain has an allocated last slot, so its bit is set; bits are per slot residence)
        uint32_t prev;
        const uint32_t at = find_entry_overflow(fp, dp.table, owned_entries(fp), local, kx, ky, kz, prev);
        return at == ~0u ? VH_FREE_BLOCK : dp.table[at].ptr;
    }
    const VoxelEntry *bucket = dp.table + (size_t)local * fp.bucketSize;
    for (uint32_t i = 0; i < fp.bucketSize; ++i) {
        const VoxelEntry e = bucket[i];
        if (e.ptr == VH_FREE_BLOCK) return VH_FREE_BLOCK;      // prefix property
        if (e.pos[0] == kx && e.pos[1] == ky && e.pos[2] == kz) return e.ptr;
    }
    return VH_FREE_BLOCK;
}

// Spec (DESIGN.md "raycast", oracle/vh_oracle.c vho_raycast): samples at camera
// depth t_i = tMin + i*voxelSize, nearest-voxel classification, first pair of
// consecutive valid samples with sdf_prev > 0 >= sdf_cur, linear interpolation.
// 16x16-pixel tiles: a wave is a 16x4 patch of neighbouring rays, which walk
// the same blocks and keep the bucket / voxel lines hot in L2.
//
// Empty space.  A sample in an absent block (or in a macro cell of 4x4x4 blocks that holds no
// block) is invalid whatever its voxel index, so whole runs of such samples are skipped -- but
// only samples that are CERTAINLY in the empty cell: the linear ray model o + d*t used for the
// bounds differs from the sample positions T*(dx*t, dy*t, t) by ~1e-6 m, and the cell is shrunk by
// kSkipMargin voxels per side (1e-2 voxel = 2e-4 m at 2 cm voxels) when it is intersected.  The
// ray jumps to the last sample certainly inside; the next sample is evaluated exactly, like the
// oracle evaluates every sample, and tells which cell comes next.
// Measured and dropped: adopting the neighbouring cell after a clear single-face
// exit without evaluating that sample (bit-equal, but the lanes of a wave then sit in different
// code paths: 47 -> 64 us), fetching a ray's next in-block voxels together, a divide-free voxel index, and the exact
// division by voxelSize through a hoisted refined reciprocal (div_fixed: 47.3 vs 47.3 us).  Half-filled waves
// (32 rays per wave, twice the waves) take 69.7 instead of 47.0 us: the kernel is bound by instruction issue
// per SIMD about as much as by its load chains, so more waves for the same rays do not pay.  Rotating the loop by one
// sample (a voxel read issued in iteration i and judged in iteration i+1, behind the next sample's arithmetic;
// bit-equal) takes 51.0 instead of 47.2 us.
constexpr float kSkipMargin = 0.01f;     // voxels

// kPatch: pixels of a wave inside the 16x16 tile: 0 = 16x4 rows, 1 = 8x8 square
template <int kPatch>
__global__ __launch_bounds__(256) void raycast_kernel(const FrameParams fp, const DevPtrs dp, float fx, float fy,
                                                      float cx, float cy, float tMin, int nSteps,
                                                      float *__restrict__ depthOut, int xcdAware)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // Workgroups are handed to the 8 XCDs round robin (workgroup b runs on XCD b % 8, each with its own
    // L2).  With xcdAware the tiles are renumbered so that an XCD gets a contiguous run of image tiles:
    // neighbouring rays, which sample the same blocks, then share an L2.
    int tx = blockIdx.x, ty = blockIdx.y;
    if (xcdAware) {
        const int n = gridDim.x * gridDim.y, b = blockIdx.y * gridDim.x + blockIdx.x;
        if ((n & 7) == 0) {
            const int r = (b & 7) * (n >> 3) + (b >> 3);
            ty = r / (int)gridDim.x;
            tx = r - ty * (int)gridDim.x;
        }
    }
    const int u = tx * 16 + (kPatch == 0 ? (int)(threadIdx.x & 15) : (wave & 1) * 8 + (lane & 7));
    const int v = ty * 16 + (kPatch == 0 ? (int)(threadIdx.x >> 4) : (wave >> 1) * 8 + (lane >> 3));
    if (u >= fp.width || v >= fp.height) return;
    const float dx = ((float)u - cx) / fx;
    const float dy = ((float)v - cy) / fy;
    const float dt = fp.voxelSize;
    const float invDt = __builtin_amdgcn_rcpf(dt) * (1.0f - 1.0e-6f);   // never over-estimates a step count
    // world-space ray per unit of camera depth (only used to bound empty-cell skips)
    const float rayD[3] = {fp.T[0] * dx + fp.T[1] * dy + fp.T[2], fp.T[4] * dx + fp.T[5] * dy + fp.T[6],
                           fp.T[8] * dx + fp.T[9] * dy + fp.T[10]};
    const float rayO[3] = {fp.T[3], fp.T[7], fp.T[11]};
    float invD[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) invD[a] = (rayD[a] != 0.0f) ? __builtin_amdgcn_rcpf(rayD[a]) : 0.0f;
    bool prevValid = false, haveKey = false, haveMacro = false, macroEmpty = false;
    float prevSdf = 0.0f, prevT = 0.0f, hit = 0.0f;
    int ckx = 0, cky = 0, ckz = 0, cptr = VH_FREE_BLOCK;
    int cmx = 0, cmy = 0, cmz = 0;
    for (int i = 0; i < nSteps; ++i) {
        const float tt = tMin + (float)i * dt;
        const float4 pw = mat4_mul(fp.T, dx * tt, dy * tt, tt, 1.0f);
        const int vx = world2voxel1(pw.x, fp.voxelSize);
        const int vy = world2voxel1(pw.y, fp.voxelSize);
        const int vz = world2voxel1(pw.z, fp.voxelSize);
        const int kx = voxel2block1(vx), ky = voxel2block1(vy), kz = voxel2block1(vz);
        if (!haveKey || kx != ckx || ky != cky || kz != ckz) {
            ckx = kx; cky = ky; ckz = kz;
            haveKey = true;
            const int mx = kx >> 2, my = ky >> 2, mz = kz >> 2;          // macro cell of 4x4x4 blocks
            if (!haveMacro || mx != cmx || my != cmy || mz != cmz) {
                cmx = mx; cmy = my; cmz = mz;
                haveMacro = true;
                const uint32_t hm = macro_hash(mx, my, mz);
                macroEmpty = !((dp.macroBits[hm >> 5] >> (hm & 31u)) & 1u);
            }
            cptr = macroEmpty ? VH_FREE_BLOCK : lookup_block(fp, dp, kx, ky, kz);
        }
        if (cptr == VH_FREE_BLOCK) {
            prevValid = false;
            float tExit = 3.0e38f;
            // an empty macro cell (no block in 4x4x4) is skipped whole: 32 voxels per side
            const int cell[3] = {macroEmpty ? cmx * 32 : kx * 8, macroEmpty ? cmy * 32 : ky * 8,
                                 macroEmpty ? cmz * 32 : kz * 8};
            const float span = macroEmpty ? 31.5f : 7.5f;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                // the cell spans voxel centres c .. c+span-0.5, i.e. world [(c-0.5)vs, (c+span)vs)
                const float lo = ((float)cell[a] - 0.5f + kSkipMargin) * fp.voxelSize;
                const float hi = ((float)cell[a] + span - kSkipMargin) * fp.voxelSize;
                // approximate reciprocals (1 ulp) are fine here: the margin absorbs them
                if (rayD[a] > 0.0f) tExit = __builtin_fminf(tExit, (hi - rayO[a]) * invD[a]);
                else if (rayD[a] < 0.0f) tExit = __builtin_fminf(tExit, (lo - rayO[a]) * invD[a]);
            }
            const float steps = (tExit - tMin) * invDt;     // last sample index at or before tExit
            if (steps > (float)i && steps < 2.0e9f) i = min((int)steps, nSteps - 1);
            continue;
        }
        // present block: classify the sample from its voxel
        const int lx = (int)((uint32_t)vx - (uint32_t)kx * 8u);
        const int ly = (int)((uint32_t)vy - (uint32_t)ky * 8u);
        const int lz = (int)((uint32_t)vz - (uint32_t)kz * 8u);
        const Voxel s = dp.blocks[(size_t)cptr + (size_t)(lz * 64 + ly * 8 + lx)];
        if (!(s.weight > 0.0f)) { prevValid = false; continue; }
        if (prevValid && prevSdf > 0.0f && s.sdf <= 0.0f) {
            hit = prevT + (dt * prevSdf) / (prevSdf - s.sdf);
            break;
        }
        prevValid = true; prevSdf = s.sdf; prevT = tt;
    }
    depthOut[(size_t)v * fp.width + u] = hit;
}


// ---------------------------------------------------------------------------
// raycast as a voxel DDA (option "raycast_mode" = VH_RAYCAST_DDA, the default)
// ---------------------------------------------------------------------------
// The traversal the reference's shader intends (raycastSDF.frag:121-177: Amanatides-Woo between the ray's two
// ends), re-specified so that it can be exact (oracle/vh_oracle.c: vho_raycast_dda, where the spec is written
// out).  In short: g(t) = G + E*t in voxel-grid units, voxel = floor(g); the crossing out of coordinate c on
// axis a happens at tnext_a(c) = ((float)c - Gs_a) * invE_a, a pure function of the integer coordinate; the three
// monotone event sequences are merged by (t, axis priority y < z < x), the shader's own choice at :156-170; every
// visited voxel of an allocated block with weight > 0 is a sample placed at the camera depth of the voxel's
// centre; first + -> - pair of consecutive valid samples = surface, linear interpolation.
//
// Because the state of the walk is (integer voxel, tnext of three integer coordinates) and the order of the
// crossing events is a total order, the state after ANY prefix of the events can be computed directly:
//   * leaving an empty cell in one step: the exit event is the first of the cell's three boundary events, every
//     other axis advances past exactly those of its events that precede it (dda_advance: a float estimate of the
//     coordinate, then the merge predicate itself decides, so the estimate's rounding never matters);
//   * starting at a later depth tau: every axis advances past its events with t < tau (dda_start).
// The image therefore does not depend on what is skipped, as long as nothing allocated is: the kernel skips on
// hashed bitmaps whose stale or colliding bits only make it skip less -- empty 4x4x4-block macro cells (32
// voxels), absent blocks (8) -- and on a conservative beam test that the 64 rays of a wave share (below).
struct DdaAxis {
    float G, E, invE, Gs;     // invE = 0: the axis never steps (|E| <= 1e-20)
    int s;                    // +1 / -1
};

__device__ __forceinline__ float dda_tnext(const DdaAxis &ax, int c)
{
    return ax.invE != 0.0f ? ((float)c - ax.Gs) * ax.invE : __builtin_inff();
}

// merge order of two events on different axes; prio: y = 0, z = 1, x = 2 (raycastSDF.frag:156-170)
__device__ __forceinline__ bool dda_before(float tb, int prioB, float ta, int prioA)
{
    return tb < ta || (tb == ta && prioB < prioA);
}

// the coordinate axis b has reached when the exit event (te, prioX) of another axis fires: the first c from
// `cur` towards `last` (inclusive) whose own crossing does not precede the exit
__device__ __forceinline__ int dda_advance(const DdaAxis &ax, int prioB, int cur, int last, float te, int prioX)
{
    if (ax.invE == 0.0f || cur == last) return cur;
    const int lo = min(cur, last), hi = max(cur, last);
    int e = f2i_rz(__builtin_floorf(ax.G + ax.E * te));
    e = min(max(e, lo), hi);
    while (e != last && dda_before(dda_tnext(ax, e), prioB, te, prioX)) e += ax.s;
    while (e != cur && !dda_before(dda_tnext(ax, e - ax.s), prioB, te, prioX)) e -= ax.s;
    return e;
}

// the coordinate of axis b once every event with t < tau has been taken, starting from `cur` (its coordinate at
// t_min): the first c from `cur` on, in the axis' direction, whose own crossing is not before tau
__device__ __forceinline__ int dda_start(const DdaAxis &ax, int cur, float tau)
{
    if (ax.invE == 0.0f) return cur;
    int e = f2i_rz(__builtin_floorf(ax.G + ax.E * tau));
    e = ax.s > 0 ? max(e, cur) : min(e, cur);
    while (dda_tnext(ax, e) < tau) e += ax.s;
    while (e != cur && !(dda_tnext(ax, e - ax.s) < tau)) e -= ax.s;
    return e;
}

// ---- the split form (option "raycast_split"): the cooperative raycast as three launches ----
// The fused cooperative kernel below is as long as its heaviest wave: a silhouette or grazing patch lists 8-13 blocks and
// walks them one after the other, ~3.5 us each, while the mean wave lists 2.7-4.2 (profiles/r04_raycast_stamps.txt).  The
// split form keeps steps 1 and 2 (beam, resolve: uniform work) in a launch of their own, which PUBLISHES every patch's
// listed blocks as items {block key, voxel pointer, patch}; a second launch walks one (patch, block) item per wave, whatever
// patch it came from, and merges the candidates of a ray with ONE 64-bit atomicMin on the ray's word -- a candidate is
// {arrival event of the hit voxel (t, axis priority), item, voxel}, the events of a ray are totally ordered, so the minimum
// IS the hit the sequential walk finds first; a third launch turns the winning word into depth (+ normal) with the fused
// kernel's arithmetic.  Queues: one per (list position, shard of 8 by workgroup index) so that no counter is hot -- a
// workgroup takes one returning atomic per list position for its four patches -- and so that the items of list position 0
// (the blocks nearest the camera) are walked first: by the time a patch's later blocks are taken, most of its rays already
// hold a candidate that precedes the block's entry event and skip it, as in the fused kernel's front-to-back order.
constexpr int kRcRanks = 32;                      // list positions with a queue of their own (a longer list: the per-lane walk)
constexpr int kRcShards = 8;
constexpr int kRcSegs = kRcRanks * kRcShards;     // 256: one uint4 of counts per lane in the item launch
constexpr unsigned long long kRcNone = ~0ull;                    // a ray's word: no candidate
constexpr uint32_t kRcDoneKey = 0xfffffffeu;                     // ... top half: the first launch has written the pixel itself (fall-back walk)
struct RcItem {
    int kx, ky, kz, ptr;          // block key, first voxel of the block
    uint32_t patch;               // linear index of the 8x8 (16x4) pixel patch
    uint32_t pad[3];
};
struct RaycastSplit {
    float *state;                 // [patch][9][64]: E[3], invE[3] (0: the axis never steps), c[3] (int bits) of every ray
    unsigned long long *best;     // [patch][64]: the ray's best candidate {ordered t : 32 | priority : 2 | item : 21 | voxel : 9}
    RcItem *items;                // [kRcSegs][segCap]
    uint32_t *counts;             // [kRcSegs]; zero between calls (the last launch clears them)
    uint32_t segCap;              // >= the patches one shard can hold
    uint32_t patchesX;            // patches per image row
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

struct RaycastArgs {
    float fx, fy, cx, cy;
    float tMin, tMax;
    float zrow[4];            // row 2 of the inverse pose, first three scaled by voxelSize: camera depth of a voxel centre
    float G[3];               // pose translation / voxelSize + 0.5: the camera centre in voxel-grid units (same for every ray)
    float invVs;              // 1 / voxelSize (beam boxes only)
    int budget;               // hang guard: more steps than any ray of this view can take (host: vh_raycast)
    int xcdAware;
    int beam;                 // 2: the cooperative form (one block list per wave); 1: per-lane walk behind a beam front end; 0: per-lane walk
                              // from t_min (A/B; views with t_min <= 0)
    unsigned long long *stamps;   // diagnostics (tools/raycast_stamps.py): per wave {start, end} of s_memrealtime (100 MHz), or null
    uint32_t stampsItemBase;      // diagnostics: first 8-word record of the item launch's waves in `stamps` (behind the list launch's)
    RaycastSplit sp;              // the split form's buffers (raycast_dda_kernel<.., kSplit>, raycast_items_kernel, raycast_resolve_kernel)
};

// voxel (vx,vy,vz) if its block is allocated and its weight > 0 (normals: the neighbours of the hit voxel)
__device__ __forceinline__ bool dda_voxel(const FrameParams &fp, const DevPtrs &dp, int vx, int vy, int vz, int kx, int ky,
                                          int kz, int cptr, float &sdf)
{
    const int bx = vx >> 3, by = vy >> 3, bz = vz >> 3;
    int ptr = cptr;
    if (bx != kx || by != ky || bz != kz) ptr = lookup_block(fp, dp, bx, by, bz);
    if (ptr == VH_FREE_BLOCK) return false;
    const Voxel s = dp.blocks[(size_t)ptr + (size_t)(((vz & 7) << 6) | ((vy & 7) << 3) | (vx & 7))];
    sdf = s.sdf;
    return s.weight > 0.0f;
}

// normal of a hit: the TSDF gradient at the hit voxel (central difference where both neighbours are samples, else one-sided,
// else no normal), normalised, rotated into the camera frame (R^T w), w = 0
__device__ __forceinline__ float4 dda_normal(const FrameParams &fp, const DevPtrs &dp, int hx, int hy, int hz, int hptr)
{
    float4 n = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const int kx = hx >> 3, ky = hy >> 3, kz = hz >> 3;
    float here = 0.0f, g[3] = {0.0f, 0.0f, 0.0f};
    bool ok = dda_voxel(fp, dp, hx, hy, hz, kx, ky, kz, hptr, here);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float sp = 0.0f, sm = 0.0f;
        const bool hp = dda_voxel(fp, dp, hx + (a == 0), hy + (a == 1), hz + (a == 2), kx, ky, kz, hptr, sp);
        const bool hm = dda_voxel(fp, dp, hx - (a == 0), hy - (a == 1), hz - (a == 2), kx, ky, kz, hptr, sm);
        if (hp && hm) g[a] = (sp - sm) * 0.5f;
        else if (hp) g[a] = sp - here;
        else if (hm) g[a] = here - sm;
        else ok = false;
    }
    if (ok) {
        const float len = __builtin_sqrtf(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
        if (len > 0.0f) {
            const float w0 = g[0] / len, w1 = g[1] / len, w2 = g[2] / len;
            n.x = fp.T[0] * w0 + fp.T[4] * w1 + fp.T[8] * w2;          // R^T * w: world -> camera
            n.y = fp.T[1] * w0 + fp.T[5] * w1 + fp.T[9] * w2;
            n.z = fp.T[2] * w0 + fp.T[6] * w1 + fp.T[10] * w2;
        }
    }
    return n;
}

// Beam front end.  The 64 rays of a wave (an 8x8 or 16x4 pixel patch) are nearly parallel and a few voxels apart,
// and each of them spends most of its look-ups on the empty space in front of the first surface.  That part of
// the march is done ONCE per wave and in parallel instead of 64 times in sequence: lane i takes slab i of the
// depth range, bounds the part of the beam (all rays of the patch) inside the slab by a box in voxel-grid units,
// and tests the cells that box touches -- one load round trip for all 64 slabs; the first slab with a set bit
// (ballot) gives a depth tau before which no ray of the wave can meet an allocated block, and every lane starts
// its own exact walk there (dda_start).  Level 1 tests macro-cell bits over the whole range, level 2 bucket bits
// in half-block slabs behind it.  Conservative by construction: the box is grown by 2 % of a voxel plus 1e-5 of
// its coordinates (the rounding of this arithmetic and of the walk's crossing times is 1e-7 of them), a box that
// spans more than two cells on an axis counts as occupied.
struct Beam {
    float dx0, dx1, dy0, dy1;             // direction bounds of the patch's rays, camera frame (z = 1)
};

// kLevel 1: macro-cell bits (cells of 32 voxels); 2: bucket bits of the blocks (8 voxels)
template <int kLevel>
__device__ __forceinline__ bool beam_slab_occupied(const FrameParams &fp, const DevPtrs &dp, const RaycastArgs &ra,
                                                   const Beam &bm, float ta, float tb)
{
    // camera-frame box of the beam between depths ta and tb
    const float xa0 = bm.dx0 * ta, xa1 = bm.dx0 * tb, xb0 = bm.dx1 * ta, xb1 = bm.dx1 * tb;
    const float ya0 = bm.dy0 * ta, ya1 = bm.dy0 * tb, yb0 = bm.dy1 * ta, yb1 = bm.dy1 * tb;
    const float lo[3] = {__builtin_fminf(__builtin_fminf(xa0, xa1), __builtin_fminf(xb0, xb1)),
                         __builtin_fminf(__builtin_fminf(ya0, ya1), __builtin_fminf(yb0, yb1)), __builtin_fminf(ta, tb)};
    const float hi[3] = {__builtin_fmaxf(__builtin_fmaxf(xa0, xa1), __builtin_fmaxf(xb0, xb1)),
                         __builtin_fmaxf(__builtin_fmaxf(ya0, ya1), __builtin_fmaxf(yb0, yb1)), __builtin_fmaxf(ta, tb)};
    const float invVs = ra.invVs;                 // (approximate is fine: the box is grown)
    constexpr int kShift = kLevel == 1 ? 5 : 3;
    int c0[3], c1[3];
    bool huge = false;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float wl = 0.0f, wh = 0.0f;                   // R_a . box, interval arithmetic
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float r = fp.T[4 * a + j] * invVs;
            const float p = r * lo[j], q = r * hi[j];
            wl += __builtin_fminf(p, q);
            wh += __builtin_fmaxf(p, q);
        }
        const float gl = ra.G[a] + wl, gh = ra.G[a] + wh;
        const float m = 0.02f + 1.0e-5f * __builtin_fmaxf(__builtin_fabsf(gl), __builtin_fabsf(gh));
        c0[a] = f2i_rz(__builtin_floorf(gl - m)) >> kShift;
        c1[a] = f2i_rz(__builtin_floorf(gh + m)) >> kShift;
        huge |= !(c1[a] - c0[a] <= 1) || !(gl == gl) || !(gh == gh);      // more than 2 cells on an axis, or NaN
    }
    if (huge) return true;
    // the (at most) 2 x 2 x 2 cells of the box: eight independent loads, one round trip (a loop over the cells with a
    // load in its body made them eight -- or, with 3 cells per axis, 27 -- round trips: 4-13 us per wave)
    uint32_t word[8], bit[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int x = (k & 1) ? c1[0] : c0[0], y = (k & 2) ? c1[1] : c0[1], z = (k & 4) ? c1[2] : c0[2];
        if (kLevel == 1) {
            const uint32_t hm = macro_hash(x, y, z);
            word[k] = dp.macroBits[hm >> 5];
            bit[k] = hm & 31u;
        } else {
            const uint32_t h = hash_block(x, y, z, fp.numBuckets);
            const bool mine = h >= fp.bucketLo && h < fp.bucketHi;
            const uint32_t local = mine ? h - fp.bucketLo : 0u;
            word[k] = mine ? dp.bucketBits[local >> 5] : 0u;
            bit[k] = local & 31u;
        }
    }
    bool any = false;
#pragma unroll
    for (int k = 0; k < 8; ++k) any |= ((word[k] >> bit[k]) & 1u) != 0u;
    return any;
}


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
#ifndef VH_COOP_K
#define VH_COOP_K 1             // voxels fetched per round trip in the block walk (with 5 waves per SIMD: 1: 35.3 us, 2: 38.1; with 4: 38.9 / 40.6;
                                // round 4: 2 only for waves whose list has 3 / 4 / 5 / 6 blocks or more, the loop built twice: 34.9 / 35.1 / 34.8 / 34.7 us
                                // against 30.4 -- 16 registers spilled instead of 2, whichever loop a wave runs)
#endif
constexpr int kCoopK = VH_COOP_K;
#ifndef VH_COOP_SUBS
#define VH_COOP_SUBS 4
#endif
constexpr int kCoopSubs = VH_COOP_SUBS;        // slabs per lane and window of the cooperative raycast (64 x this many half-block slabs)
#ifndef VH_COOP_LDS
#define VH_COOP_LDS 0      // 1: the walked block is staged in LDS (4 KiB per wave, one coalesced round trip), 0: its voxels are gathered (35.0 vs 32.4 us: 19 registers spilled instead of 8)
#endif
#ifndef VH_COOP_PRIO
#define VH_COOP_PRIO 1
#endif
#ifndef VH_COOP_PREFETCH
#define VH_COOP_PREFETCH 0   // 1: the walked blocks' lines are requested one block ahead (below; measured: 31.8-32.2 us against 31.1-31.2 without)
#endif
#ifndef VH_COOP_RESOLVE
#define VH_COOP_RESOLVE 1    // 1: the set's cells are looked up eight lanes per bucket (0: one lane per cell, slot after slot)
#endif
#ifndef VH_DDA_BLOCK_WAVES
#define VH_DDA_BLOCK_WAVES 4    // waves (pixel patches) per workgroup: 4 = a 16x16 tile, 1 = a patch of its own
#endif
constexpr int kDdaBlockWaves = VH_DDA_BLOCK_WAVES;
constexpr int kCoopSlots = 256;                       // per wave: cells with a set bucket bit
constexpr uint32_t kCoopUnresolved = 0x7ffffffeu;     // sPtr: not looked up yet
struct CoopShared {
    uint32_t tag[kDdaBlockWaves][kCoopSlots];
    uint32_t ptr[kDdaBlockWaves][kCoopSlots];
    uint32_t list[kDdaBlockWaves][kCoopSlots];     // allocated cells: slot | depth key << 16 (the list is walked front to back)
    uint16_t cells[kDdaBlockWaves][kCoopSlots];    // the set's occupied slots in order of insertion (what step 2 resolves)
    uint32_t count[kDdaBlockWaves];
    uint32_t nItems[kDdaBlockWaves];               // split form: the patches' list lengths, and where each list position's items go
    uint32_t segBase[kRcRanks];
#if VH_COOP_LDS
    Voxel block[kDdaBlockWaves][kBlockVoxels];     // the block the wave is walking (4 KiB per wave)
#endif
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

// Shape of the kernel.  A ray's work is small (C2: ~25 absent blocks stepped over, 1.5 allocated blocks, ~10 voxels)
// but what was measured on the way here (per-wave timeline, tools/raycast_stamps.py; counters, profiles/) is that the
// launch is as long as its SLOWEST wave -- silhouette and grazing patches -- and that a wave advances at ~4 cycles per
// instruction whatever its neighbours do: a per-lane loop with exact cell exits took 67 us on C2, two-phase
// (skip / walk) loops with 8-voxel chunks 73-81, the block-level DDA 72, chunks of 2 cells 60 (4: 70, 8: 89), the
// fixed-step march 47.  What made the difference was to stop doing the search 64 times per wave: the cooperative form
// (above; the default) at 46 us.  The per-lane walk below remains as its fall-back and as raycast_beam 1 / 0:
//   * behind a beam front end (beam = 1) or from t_min (0), one loop serves two levels: blocks (sh = 3) through
//     absent space, voxels (sh = 0) inside allocated blocks; both are the same merge of three monotone crossing-time
//     sequences (a block's crossing is the voxel event out of its last coordinate), so no lane waits for another
//     lane's phase; kDdaK cells ahead are enumerated by arithmetic alone, their loads issued together, then judged
//     in order; voxel coordinates are rebuilt only when an allocated block is entered from an absent one.
#ifndef VH_DDA_K
#define VH_DDA_K 2
#endif
#ifndef VH_DDA_PRIO
#define VH_DDA_PRIO 0       // 1: a wave that is still walking after 12 / 24 / 40 rounds raises its priority (measured: no gain)
#endif
constexpr int kDdaK = VH_DDA_K;
#ifndef VH_DDA_WAVES
#define VH_DDA_WAVES 5      // waves per SIMD the register budget must allow.  A 640x480 view is 4 800 waves = 4.7 per SIMD: with 4 resident the
                            // last 704 start when the first ones end (12-20 us into a 40 us launch); with 5 (96 registers, 6-9 of them spilled
                            // once the front end had been slimmed) all start at once: 40.6 -> 35.3 us.  6 (80 registers, 30-60 spilled): slower.
#endif

template <int kPatch, bool kNormals, bool kSplit = false>
__global__ __launch_bounds__(64 * kDdaBlockWaves, VH_DDA_WAVES) void raycast_dda_kernel(const FrameParams fp, const DevPtrs dp, const RaycastArgs ra,
                                                          float *__restrict__ depthOut, float4 *__restrict__ normalOut)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long stamp0 = ra.stamps ? __builtin_amdgcn_s_memrealtime() : 0ull;
    int tx = blockIdx.x, ty = blockIdx.y;
    if (ra.xcdAware) {                       // each XCD (own L2) renders a contiguous run of image tiles
        const int n = gridDim.x * gridDim.y, b = blockIdx.y * gridDim.x + blockIdx.x;
        if ((n & 7) == 0) {
            const int r = (b & 7) * (n >> 3) + (b >> 3);
            ty = r / (int)gridDim.x;
            tx = r - ty * (int)gridDim.x;
        }
    }
    // the wave's pixel patch
    const int pu = kDdaBlockWaves == 1 ? tx * (kPatch == 0 ? 16 : 8) : tx * 16 + (kPatch == 0 ? 0 : (wave & 1) * 8);
    const int pv = kDdaBlockWaves == 1 ? ty * (kPatch == 0 ? 4 : 8) : ty * 16 + (kPatch == 0 ? wave * 4 : (wave >> 1) * 8);
    const int u = pu + (kPatch == 0 ? (lane & 15) : (lane & 7));
    const int v = pv + (kPatch == 0 ? (lane >> 4) : (lane >> 3));
    const bool inImage = u < fp.width && v < fp.height;
    const float dx = ((float)u - ra.cx) / ra.fx;
    const float dy = ((float)v - ra.cy) / ra.fy;
    const float vs = fp.voxelSize;
    DdaAxis ax[3];
    int c[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float D = fp.T[4 * a + 0] * dx + fp.T[4 * a + 1] * dy + fp.T[4 * a + 2];
        ax[a].G = ra.G[a];
        ax[a].E = D / vs;
        const bool active = __builtin_fabsf(ax[a].E) > 1.0e-20f;
        ax[a].invE = active ? 1.0f / ax[a].E : 0.0f;
        ax[a].s = ax[a].E > 0.0f ? 1 : -1;
        ax[a].Gs = ax[a].E > 0.0f ? ax[a].G - 1.0f : ax[a].G;
        c[a] = f2i_rz(__builtin_floorf(ax[a].G + ax[a].E * ra.tMin));
    }
    bool live = inImage;
    bool found = false;
    float hit = 0.0f;
    int hx = 0, hy = 0, hz = 0, hptr = VH_FREE_BLOCK;          // (per-lane walk: the last valid sample's voxel;) after a hit: the hit voxel and its block
    const int prio[3] = {2, 0, 1};
    bool coopDone = false;
    unsigned long long stampP1 = 0ull;                        // diagnostics: the ray set-up is done
    unsigned long long stampA = stamp0, stampB = stamp0;      // diagnostics: the set is built / the list is resolved
    int coopList = 0, coopWalks = 0;
#ifdef VH_RAYCAST_DIAG
    unsigned long long diagEntry = 0ull, diagWalk = 0ull;     // diagnostics build: time in the entry tests / in the voxel loops, and the loops' rounds
    int diagRounds = 0;
#endif
    if (ra.beam == 2) {
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
            const float e0 = __shfl(ax[a].E, 0), e1 = __shfl(ax[a].E, kPatch == 0 ? 15 : 7);
            const float e2 = __shfl(ax[a].E, kPatch == 0 ? 48 : 56), e3 = __shfl(ax[a].E, 63);
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
#if VH_COOP_RESOLVE
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
#endif
            for (int cb = cellBegin; cb < nCells; cb += 64) {
                const bool todo = cb + lane < nCells;
                const int slot = todo ? (int)cells[cb + lane] : 0;
                const uint32_t t = tags[slot] - 1u;
                const int qx = base0 + (int)(t & 1023u), qy = base1 + (int)((t >> 10) & 1023u), qz = base2 + (int)(t >> 20);
#if VH_COOP_RESOLVE
                const int ptr = todo ? (int)ptrs[slot] : VH_FREE_BLOCK;
#else
                const uint32_t myLocal = hash_block(qx, qy, qz, fp.numBuckets) - fp.bucketLo;      // (a set bit: the bucket is this shard's)
                int ptr = VH_FREE_BLOCK;
                if (todo) {
                    if (fp.flags & kFlagOverflow) {
                        uint32_t prev;
                        const uint32_t at = find_entry_overflow(fp, dp.table, owned_entries(fp), myLocal, qx, qy, qz, prev);
                        if (at != ~0u) ptr = dp.table[at].ptr;
                    } else {
                        const VoxelEntry *bucket = dp.table + (size_t)myLocal * fp.bucketSize;
                        for (uint32_t i = 0; i < fp.bucketSize; ++i) {        // getVoxelEntry4Block, VoxelUtils.cu:362-382
                            const VoxelEntry e = bucket[i];
                            if (e.ptr == VH_FREE_BLOCK) break;                // prefix property: a free slot ends the bucket
                            if (e.pos[0] == qx && e.pos[1] == qy && e.pos[2] == qz) { ptr = e.ptr; break; }
                        }
                    }
                    ptrs[slot] = (uint32_t)ptr;
                }
#endif
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
            if constexpr (kSplit) continue;             // (the walk is the item launch's; every window is listed)
#if VH_COOP_PRIO
            // The launch is as long as its slowest wave, and the slowest waves are the ones with the longest lists: they get the
            // issue slots first (s_setprio; 35.6 -> 32.6 us; thresholds 6/4/3, 7/5/3 and 10/7/5 measured the same).
            {
                const int n = nList - listBegin;
                if (n >= 8) __builtin_amdgcn_s_setprio(3); else if (n >= 6) __builtin_amdgcn_s_setprio(2); else if (n >= 4) __builtin_amdgcn_s_setprio(1);
            }
#endif
            // ---- 3. every ray against every new block of the list ----
            // (a wave-uniform loop: the block's key and voxel pointer are scalars.  Measured alternative: every ray walking
            // its OWN blocks, one per round -- the busiest ray of a wave enters as many blocks as the wave walks, 2.1 vs 2.2
            // rounds, and the per-lane block pointer made it 44.7 us against 40.2)
            {
#if VH_COOP_PREFETCH
                // The voxel loop below is a chain of dependent gathers, and a block's lines are met for the first time by its
                // first rounds (the pool is far larger than an L2; diagnostics build: ~690 cycles per round for 55 instructions).
                // So every lane touches one 64-byte piece of the NEXT block of the list -- the block's whole 4 KiB, one request per
                // lane -- while this block's entry arithmetic and walk run; the value is never used, only waited for a block later.
                uint32_t pf = 0u;
                if (listBegin < nList) {
                    const int s0_ = __builtin_amdgcn_readfirstlane((int)(list[listBegin] & 0xffffu));
                    const int p0_ = __builtin_amdgcn_readfirstlane((int)ptrs[s0_]);
                    pf = reinterpret_cast<const uint32_t *>(dp.blocks + (size_t)p0_)[lane * 16];
                }
#endif
                for (int i = listBegin; i < nList; ++i) {
                    if (__ballot(!final_) == 0ull) break;
#if VH_COOP_PREFETCH
                    asm volatile("" ::"v"(pf));
                    if (i + 1 < nList) {
                        const int sN = __builtin_amdgcn_readfirstlane((int)(list[i + 1] & 0xffffu));
                        const int pN = __builtin_amdgcn_readfirstlane((int)ptrs[sN]);
                        pf = reinterpret_cast<const uint32_t *>(dp.blocks + (size_t)pN)[lane * 16];
                    }
#endif
                    // (measured in round 4: {tag, pointer} kept in walking order beside the list, one LDS round trip here instead of three
                    // dependent ones: 31.5 against 31.0 us -- 5 registers spilled instead of 2)
                    const int slot = __builtin_amdgcn_readfirstlane((int)(list[i] & 0xffffu));
                    const uint32_t tg = (uint32_t)__builtin_amdgcn_readfirstlane((int)tags[slot]) - 1u;
                    const int bptr = __builtin_amdgcn_readfirstlane((int)ptrs[slot]);
                    const int kk[3] = {base0 + (int)(tg & 1023u), base1 + (int)((tg >> 10) & 1023u), base2 + (int)(tg >> 20)};
#ifdef VH_RAYCAST_DIAG
                    const unsigned long long dg0 = __builtin_amdgcn_s_memrealtime();
#endif
#if VH_COOP_LDS == 2
                    // LDS-DMA of the block's 4 KiB (four global_load_lds_dwordx4, no registers), issued BEFORE the entry test: its
                    // round trip runs under the ~0.8 us of entry arithmetic, and every step of the walk then reads at LDS latency
                    // instead of waiting ~690 cycles for a gather.  One buffer per wave: the previous block's walk is over (the
                    // lanes have met again), a DMA that was never waited for (no lane entered its block) is waited for here.
                    {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __builtin_amdgcn_wave_barrier();
                        const char *src = reinterpret_cast<const char *>(dp.blocks + (size_t)bptr) + lane * 16;
                        char *dst = reinterpret_cast<char *>(sh_.block[wave]);
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + j * 1024),
                                                             (__attribute__((address_space(3))) void *)(dst + j * 1024), 16, 0, 0);
                    }
#endif
                    const CoopEntry e = coop_entry(ax, c, kk, ra.tMax);
                    const float tE = e.tE;
                    const int pE = e.pE, xe = e.xe;
                    const bool inside = e.inside;
                    const bool enters = e.enters && !final_ && dda_before(tE, pE, bestT, bestP);      // (not behind the best candidate so far)
                    if (__ballot(enters) == 0ull) continue;
                    ++coopWalks;
#if VH_COOP_LDS == 2
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the block is in LDS
                    __builtin_amdgcn_wave_barrier();
#elif VH_COOP_LDS
                    // the block's 4 KiB into LDS, 64 bytes per lane: one coalesced round trip, after which every step of every
                    // ray reads at LDS latency instead of waiting for a gather
                    {
                        __builtin_amdgcn_wave_barrier();               // (the previous block's walks are done with the buffer)
                        const float4 *src4 = reinterpret_cast<const float4 *>(dp.blocks + (size_t)bptr) + lane * 4;
                        float4 *dst4 = reinterpret_cast<float4 *>(sh_.block[wave]) + lane * 4;
                        const float4 v0 = src4[0], v1 = src4[1], v2 = src4[2], v3 = src4[3];
                        dst4[0] = v0; dst4[1] = v1; dst4[2] = v2; dst4[3] = v3;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    }
#endif
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
#if VH_COOP_LDS
                    const Voxel *blk = sh_.block[wave];
#else
                    const Voxel *blk = dp.blocks + (size_t)bptr;
#endif
#ifdef VH_RAYCAST_DIAG
                    const unsigned long long dg1 = __builtin_amdgcn_s_memrealtime();
                    diagEntry += dg1 - dg0;
#endif
#ifdef VH_RAYCAST_DIAG
                    while (__ballot(walking) != 0ull) {
                        ++diagRounds;
                        if (!walking) continue;
#else
                    while (walking) {
#endif
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
#ifdef VH_RAYCAST_DIAG
                    diagWalk += __builtin_amdgcn_s_memrealtime() - dg1;
#endif
                }
            }
            // a candidate that arrived before this window's end cannot be beaten by a block found later
            final_ = final_ || bestT < tw + window;
            if (__ballot(!final_) == 0ull) break;
        }
        if constexpr (kSplit) {
            // ---- publication: the patch's listed blocks become items of the queues, its rays' set-up goes to memory ----
            // One returning atomic per list position and WORKGROUP (lane r of wave 0 adds the number of the workgroup's patches
            // that have an r-th block to the counter of queue (r, shard)); every wave then writes its own items, lane r the r-th.
            const bool publish = !fail && nList <= kRcRanks && __ballot(inImage) != 0ull;     // (a patch outside the image: nothing to walk)
            if (lane == 0) sh_.nItems[wave] = publish ? (uint32_t)nList : 0u;
            __syncthreads();
            uint32_t cnt = 0u, before = 0u;
#pragma unroll
            for (int w = 0; w < kDdaBlockWaves; ++w) {
                const uint32_t has = sh_.nItems[w] > (uint32_t)lane ? 1u : 0u;
                cnt += has;
                before += w < wave ? has : 0u;
            }
            const uint32_t shard = (uint32_t)(blockIdx.y * gridDim.x + blockIdx.x) & (kRcShards - 1);
            if (wave == 0 && lane < kRcRanks && cnt) sh_.segBase[lane] = atomicAdd(&ra.sp.counts[lane * kRcShards + shard], cnt);
            __syncthreads();
            if (publish) {
                const uint32_t patch = (uint32_t)(pv / (kPatch == 0 ? 4 : 8)) * ra.sp.patchesX + (uint32_t)(pu / (kPatch == 0 ? 16 : 8));
                if (lane < nList) {
                    const int slot = (int)(list[lane] & 0xffffu);
                    const uint32_t tg = tags[slot] - 1u;
                    RcItem item;
                    item.kx = base0 + (int)(tg & 1023u); item.ky = base1 + (int)((tg >> 10) & 1023u); item.kz = base2 + (int)(tg >> 20);
                    item.ptr = (int)ptrs[slot];
                    item.patch = patch;
                    item.pad[0] = item.pad[1] = item.pad[2] = 0u;
                    const uint32_t at = sh_.segBase[lane] + before;        // (< segCap: a shard's queue has room for every patch of the shard)
                    ra.sp.items[(size_t)(lane * kRcShards + (int)shard) * ra.sp.segCap + at] = item;
                }
                float *st = ra.sp.state + (size_t)patch * (9 * 64) + lane;
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    st[64 * a] = ax[a].E;
                    st[64 * (3 + a)] = ax[a].invE;
                    st[64 * (6 + a)] = __int_as_float(c[a]);
                }
                ra.sp.best[(size_t)patch * 64 + lane] = kRcNone;
                if (ra.stamps && lane == 0) {
                    const size_t w = ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * kDdaBlockWaves + wave) * 8;
                    ra.stamps[w] = stamp0; ra.stamps[w + 1] = __builtin_amdgcn_s_memrealtime();
                    ra.stamps[w + 2] = stampP1 - stamp0; ra.stamps[w + 3] = (unsigned long long)(pu | (pv << 16));
                    ra.stamps[w + 4] = stampA - stamp0; ra.stamps[w + 5] = stampB - stamp0; ra.stamps[w + 6] = (unsigned long long)nList; ra.stamps[w + 7] = 0ull;
                }
                return;                                  // (the item launch walks, the resolve launch writes the pixels)
            }
            fail = true;                                 // the per-lane walk below, pixels written here
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
    }
    // ---- beam front end: the depth before which no ray of this wave can meet an allocated block ----
    if (ra.beam && !coopDone) {
        Beam bm;
        // (the corner rays' directions are those of the patch's corner lanes: no division here)
        const float a0 = __shfl(dx, 0), a1 = __shfl(dx, kPatch == 0 ? 15 : 7);
        const float b0 = __shfl(dy, 0), b1 = __shfl(dy, kPatch == 0 ? 48 : 56);
        bm.dx0 = __builtin_fminf(a0, a1); bm.dx1 = __builtin_fmaxf(a0, a1);
        bm.dy0 = __builtin_fminf(b0, b1); bm.dy1 = __builtin_fmaxf(b0, b1);
        float tau = ra.tMin;
        const float range = ra.tMax - ra.tMin;
        const float dt2 = 4.0f * vs;                                   // level 2: half-block slabs
        if (64.0f * dt2 < range) {                                      // level 1 pays when level 2 cannot span the range
            const float dt1 = range * (1.0f / 64.0f);
            const float ta = ra.tMin + (float)lane * dt1;
            const bool occ = beam_slab_occupied<1>(fp, dp, ra, bm, ta - 1.0e-4f * dt1, ta + 1.0001f * dt1);
            const unsigned long long m = __ballot(occ);
            if (m == 0ull) live = false;                                // no macro cell with a block along any ray
            else tau = ra.tMin + (float)(__ffsll((long long)m) - 1) * dt1;
        }
        if (__ballot(live) != 0ull) {
            const float ta = tau + (float)lane * dt2;
            const bool occ = ta < ra.tMax && beam_slab_occupied<2>(fp, dp, ra, bm, ta - 1.0e-4f * dt2, ta + 1.0001f * dt2);
            const unsigned long long m = __ballot(occ);
            const float t2 = tau + (m == 0ull ? 64.0f : (float)(__ffsll((long long)m) - 1)) * dt2;
            if (!(t2 < ra.tMax)) live = false;                          // nothing allocated before the rays end
            else tau = t2;
        }
        if (live && tau > ra.tMin) {
#pragma unroll
            for (int a = 0; a < 3; ++a) c[a] = dda_start(ax[a], c[a], tau);
        }
    }
    const unsigned long long stamp1 = ra.stamps ? __builtin_amdgcn_s_memrealtime() : 0ull;
    // ---- the walk: one cell per iteration, at block level (sh = 3) through absent blocks, at voxel level (sh = 0)
    // inside allocated ones.  Both levels are the same merge of three monotone crossing-time sequences -- the
    // crossing out of block coordinate k is the voxel-level event out of the block's last coordinate, so the block
    // events are a subsequence of the voxel events and merging them in the same order visits exactly the blocks the
    // voxel walk visits -- and one instruction sequence serves both: no lane waits for another lane's phase.  Voxel
    // coordinates are rebuilt only when an allocated block is entered from an absent one: on the entry axis the
    // block's first coordinate, on the others the first coordinate of the block whose own crossing is not before
    // the entry event (dda_advance).
    // The loop body is written as straight-line selects with ONE rare branch (a hit, or a bucket bit that is set):
    // on this chip a wave's scalar instructions -- mask logic and the exec-mask bookkeeping of every divergent
    // branch -- cost as much issue time as its vector instructions (one scalar unit per CU for four SIMDs), and the
    // branchy form of this loop spent 3 000 scalar against 2 900 vector instructions per wave.  The voxel of an
    // allocated block and the bucket-bit word of a block are fetched by the SAME load instruction (address select;
    // the bitmap has a word of padding).
    int prevValid = 0;
    float prevSdf = 0.0f;
    int cptr = 0, kx = 0, ky = 0, kz = 0;                      // the allocated block the ray stands in (voxel level)
    int budget = ra.budget;
    int sh = 3;                                                // level: 3 = blocks, 0 = voxels
    int q0 = c[0] >> 3, q1 = c[1] >> 3, q2 = c[2] >> 3;        // the cell at that level (voxel2Block for two's complement ints)
    int c0 = c[0], c1 = c[1], c2 = c[2];                       // block level with haveVoxel: the exact voxel the ray stands in
    int haveVoxel = 1;
    float entryT = 0.0f;                                       // block level, !haveVoxel: the event that entered the block
    int entryX = 0;
    // level-generic crossing time: out of cell q on axis a = voxel event out of (q << sh) + (s > 0 ? 2^sh - 1 : 0)
    const int s0 = ax[0].s, s1 = ax[1].s, s2 = ax[2].s;
    const int o0 = s0 > 0 ? 7 : 0, o1 = s1 > 0 ? 7 : 0, o2 = s2 > 0 ? 7 : 0;
    const float ie0 = ax[0].invE != 0.0f ? ax[0].invE : __builtin_inff(), gs0 = ax[0].invE != 0.0f ? ax[0].Gs : -1.0e30f;
    const float ie1 = ax[1].invE != 0.0f ? ax[1].invE : __builtin_inff(), gs1 = ax[1].invE != 0.0f ? ax[1].Gs : -1.0e30f;
    const float ie2 = ax[2].invE != 0.0f ? ax[2].invE : __builtin_inff(), gs2 = ax[2].invE != 0.0f ? ax[2].Gs : -1.0e30f;
#define VH_DDA_TN(q, o, gs, ie) (((float)(((q) << sh) + (sh ? (o) : 0)) - (gs)) * (ie))
    float tn0 = VH_DDA_TN(q0, o0, gs0, ie0), tn1 = VH_DDA_TN(q1, o1, gs1, ie1), tn2 = VH_DDA_TN(q2, o2, gs2, ie2);
    const bool pow2 = (fp.numBuckets & (fp.numBuckets - 1u)) == 0u;
    // kDdaK cells ahead per iteration: the walk does not depend on what the cells hold, so the next kDdaK cells at
    // the current level are enumerated by arithmetic alone, their kDdaK loads issued together (one memory round trip
    // instead of kDdaK: a wave's 64 lanes gather 64 different cache lines per load, ~500 cycles each time), then judged
    // in order.  A chunk ends early where the level changes (the ray leaves its block) or the ray ends; an allocated
    // block found among the candidates discards the steps enumerated behind it.
    int round = 0;
    while (live) {
        // The launch is as long as its slowest wave (per-wave timeline, tools/raycast_stamps.py: mean 25 us, slowest 70),
        // and while the SIMDs are full every wave gets a fifth of the issue slots: a wave that is still walking after
        // many rounds is one of the long ones (silhouette and grazing patches) and moves ahead of the short ones.
        ++round;
        if (VH_DDA_PRIO) {
            if (round == 12) __builtin_amdgcn_s_setprio(1);
            else if (round == 24) __builtin_amdgcn_s_setprio(2);
            else if (round == 40) __builtin_amdgcn_s_setprio(3);
        }
        const bool isV = sh == 0;
        int cq0[kDdaK], cq1[kDdaK], cq2[kDdaK], cx[kDdaK];
        float ct[kDdaK];
        uint32_t cl[kDdaK];
        uint2 cw[kDdaK];
        int n = 0;
        bool more = true, ends = false, left = false;
        float eT = entryT;
        int eX = haveVoxel ? -1 : entryX;
#pragma unroll
        for (int j = 0; j < kDdaK; ++j) {
            if (more) {
                cq0[j] = q0; cq1[j] = q1; cq2[j] = q2; ct[j] = eT; cx[j] = eX;
                const uint32_t hh = ((uint32_t)q0 * 73856093u) ^ ((uint32_t)q1 * 19349669u) ^ ((uint32_t)q2 * 83492791u);   // calculateHash, VoxelUtils.cu:250-259
                const uint32_t h = pow2 ? (hh & (fp.numBuckets - 1u)) : (hh % fp.numBuckets);
                const bool mine = h >= fp.bucketLo && h < fp.bucketHi;
                const uint32_t local = mine ? h - fp.bucketLo : 0u;
                cl[j] = mine ? local : ~0u;
                const uint32_t lin = (uint32_t)(((q2 & 7) << 6) | ((q1 & 7) << 3) | (q0 & 7));
                const char *addr = isV ? reinterpret_cast<const char *>(dp.blocks + ((size_t)cptr + lin))
                                       : reinterpret_cast<const char *>(dp.bucketBits + (local >> 5));
                cw[j] = *reinterpret_cast<const uint2 *>(addr);
                n = j + 1;
                // the crossing that ends this cell (raycastSDF.frag:156-170): x only when strictly first, z before x on a
                // tie, y before both
                const bool m0 = tn0 < tn1 && tn0 < tn2;
                const bool m2 = !m0 && tn2 < tn1;
                const bool m1 = !m0 && !m2;
                const float tOut = m0 ? tn0 : m2 ? tn2 : tn1;
                ends = !(tOut < ra.tMax);
                q0 += m0 ? s0 : 0; q1 += m1 ? s1 : 0; q2 += m2 ? s2 : 0;
                left = isV && ((((q0 >> 3) ^ kx) | ((q1 >> 3) ^ ky) | ((q2 >> 3) ^ kz)) != 0);
                eT = tOut; eX = m0 ? 0 : m2 ? 2 : 1;
                more = !ends && !left;
                if (more) { tn0 = VH_DDA_TN(q0, o0, gs0, ie0); tn1 = VH_DDA_TN(q1, o1, gs1, ie1); tn2 = VH_DDA_TN(q2, o2, gs2, ie2); }
            }
        }
        budget -= n;
        // ---- judged in order ----
        int ev = -1;                         // first candidate that is a hit (voxel level) / whose bucket bit is set (block level)
#pragma unroll
        for (int j = 0; j < kDdaK; ++j) {
            if (j < n && ev < 0) {
                const float sdf = __uint_as_float(cw[j].x), wgt = __uint_as_float(cw[j].y);
                const bool valid = isV && wgt > 0.0f;
                const bool isHit = valid && prevValid && prevSdf > 0.0f && sdf <= 0.0f;
                const bool bitSet = !isV && cl[j] != ~0u && ((cw[j].x >> (cl[j] & 31u)) & 1u);
                if (isHit || bitSet) {
                    ev = j;
                } else {
                    prevValid = valid ? 1 : 0;                       // (an absent block: no valid sample)
                    prevSdf = valid ? sdf : prevSdf;
                    hx = valid ? cq0[j] : hx; hy = valid ? cq1[j] : hy; hz = valid ? cq2[j] : hz;
                }
            }
        }
        if (ev >= 0) {
            int e0 = cq0[0], e1 = cq1[0], e2 = cq2[0], eX2 = cx[0];
            float eT2 = ct[0], eSdf = __uint_as_float(cw[0].x);
            uint32_t eL = cl[0];
#pragma unroll
            for (int j = 1; j < kDdaK; ++j)
                if (ev == j) { e0 = cq0[j]; e1 = cq1[j]; e2 = cq2[j]; eX2 = cx[j]; eT2 = ct[j]; eSdf = __uint_as_float(cw[j].x); eL = cl[j]; }
            if (isV) {
                // the samples sit at their voxels' centres: camera depth = row 2 of the inverse pose
                const float tc = ((ra.zrow[0] * (float)e0 + ra.zrow[1] * (float)e1) + ra.zrow[2] * (float)e2) + ra.zrow[3];
                const float tp = ((ra.zrow[0] * (float)hx + ra.zrow[1] * (float)hy) + ra.zrow[2] * (float)hz) + ra.zrow[3];
                hit = tp + ((tc - tp) * prevSdf) / (prevSdf - eSdf);
                found = true;
                hx = e0; hy = e1; hz = e2; hptr = cptr;
                break;
            }
            int ptr = VH_FREE_BLOCK;
            if (fp.flags & kFlagOverflow) {
                uint32_t prev;
                const uint32_t at = find_entry_overflow(fp, dp.table, owned_entries(fp), eL, e0, e1, e2, prev);
                if (at != ~0u) ptr = dp.table[at].ptr;
            } else {
                const VoxelEntry *bucket = dp.table + (size_t)eL * fp.bucketSize;
                for (uint32_t i = 0; i < fp.bucketSize; ++i) {            // getVoxelEntry4Block, VoxelUtils.cu:362-382
                    const VoxelEntry e = bucket[i];
                    if (e.ptr == VH_FREE_BLOCK) break;                    // prefix property
                    if (e.pos[0] == e0 && e.pos[1] == e1 && e.pos[2] == e2) { ptr = e.ptr; break; }
                }
            }
            // the steps enumerated behind candidate ev are dropped either way: the walk resumes AT it
            if (ptr != VH_FREE_BLOCK) {
                // down to voxel level: where the ray stands inside the block
                kx = e0; ky = e1; kz = e2;
                cptr = ptr;
                if (eX2 >= 0) {
                    prevValid = 0;                                  // (absent blocks lie behind: entered from outside)
                    const int pX = eX2 == 0 ? 2 : eX2 == 1 ? 0 : 1;
                    const int gk[3] = {e0, e1, e2};
                    int cc[3] = {c0, c1, c2};
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        const int nearC = (gk[a] << 3) + (ax[a].s > 0 ? 0 : 7), farC = (gk[a] << 3) + (ax[a].s > 0 ? 7 : 0);
                        cc[a] = (a == eX2) ? nearC : (ax[a].invE == 0.0f ? c[a] : dda_advance(ax[a], prio[a], nearC, farC, eT2, pX));
                    }
                    c0 = cc[0]; c1 = cc[1]; c2 = cc[2];
                }
                q0 = c0; q1 = c1; q2 = c2;
                sh = 0;
                haveVoxel = 1;
            } else {
                // a bucket that holds other keys: the block is absent; resume the block walk behind it
                prevValid = 0;
                q0 = e0; q1 = e1; q2 = e2;
                tn0 = VH_DDA_TN(q0, o0, gs0, ie0); tn1 = VH_DDA_TN(q1, o1, gs1, ie1); tn2 = VH_DDA_TN(q2, o2, gs2, ie2);
                const bool m0 = tn0 < tn1 && tn0 < tn2;
                const bool m2 = !m0 && tn2 < tn1;
                const bool m1 = !m0 && !m2;
                const float tOut = m0 ? tn0 : m2 ? tn2 : tn1;
                if (!(tOut < ra.tMax)) break;
                q0 += m0 ? s0 : 0; q1 += m1 ? s1 : 0; q2 += m2 ? s2 : 0;
                haveVoxel = 0; entryT = tOut; entryX = m0 ? 0 : m2 ? 2 : 1;
            }
            tn0 = VH_DDA_TN(q0, o0, gs0, ie0); tn1 = VH_DDA_TN(q1, o1, gs1, ie1); tn2 = VH_DDA_TN(q2, o2, gs2, ie2);
            continue;
        }
        if (ends || budget < 0) break;
        // voxel level: left the block?  Then up to block level, the exact voxel kept in case the next block is allocated too
        c0 = left ? q0 : c0; c1 = left ? q1 : c1; c2 = left ? q2 : c2;
        q0 = left ? q0 >> 3 : q0; q1 = left ? q1 >> 3 : q1; q2 = left ? q2 >> 3 : q2;
        sh = left ? 3 : sh;
        haveVoxel = isV ? 1 : 0;
        entryT = isV ? entryT : eT;
        entryX = isV ? entryX : eX;
        if (left) { tn0 = VH_DDA_TN(q0, o0, gs0, ie0); tn1 = VH_DDA_TN(q1, o1, gs1, ie1); tn2 = VH_DDA_TN(q2, o2, gs2, ie2); }
    }
#undef VH_DDA_TN
    if (ra.stamps && lane == 0) {
        const size_t w = ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * kDdaBlockWaves + wave) * 8;
        ra.stamps[w] = stamp0; ra.stamps[w + 1] = __builtin_amdgcn_s_memrealtime();
        ra.stamps[w + 2] = coopDone ? (stampP1 - stamp0) : (unsigned long long)(ra.budget - budget) | ((unsigned long long)round << 32); ra.stamps[w + 3] = (unsigned long long)(pu | (pv << 16)) | ((stamp1 - stamp0) << 32);
        ra.stamps[w + 4] = stampA - stamp0; ra.stamps[w + 5] = stampB - stamp0; ra.stamps[w + 6] = (unsigned long long)coopList; ra.stamps[w + 7] = (unsigned long long)coopWalks;
#ifdef VH_RAYCAST_DIAG
        ra.stamps[w + 6] |= (unsigned long long)diagRounds << 32;
        ra.stamps[w + 4] |= diagEntry << 32; ra.stamps[w + 5] |= diagWalk << 32;
#endif
    }
    if (!inImage) return;
    if constexpr (kSplit) {                      // a patch that took the per-lane walk: its pixels are final, the resolve launch leaves them alone
        const uint32_t patch = (uint32_t)(pv / (kPatch == 0 ? 4 : 8)) * ra.sp.patchesX + (uint32_t)(pu / (kPatch == 0 ? 16 : 8));
        ra.sp.best[(size_t)patch * 64 + lane] = (unsigned long long)kRcDoneKey << 32;
    }
    depthOut[(size_t)v * fp.width + u] = hit;
    if (!kNormals) return;
    // ---- normal of the hit: TSDF gradient at the hit voxel, normalised, camera frame, w = 0 ----
    float4 n = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (found) n = dda_normal(fp, dp, hx, hy, hz, hptr);
    normalOut[(size_t)v * fp.width + u] = n;
}

// ---------------------------------------------------------------------------
// The split form, second launch: one (patch, block) item per wave
// ---------------------------------------------------------------------------
// Step 3 of the cooperative form for ONE listed block: whether and where each ray of the item's patch enters the block, the
// voxel walk inside it, the block's first + -> - pair as the ray's candidate.  Nothing here knows what the other blocks of
// the patch yield: the candidate is merged into the ray's word with a 64-bit atomicMin (64 lanes x 8 contiguous bytes), and
// the word read beforehand only serves to skip a block the ray enters behind a candidate it already holds.  The grid is a
// tuning parameter (a grid-stride loop over the items, queue after queue: list position 0 of every shard, then 1, ...); the
// number of items is read from the queues' counters, 256 of them = one uint4 per lane + one wave scan.
#ifndef VH_ITEM_WAVES
#define VH_ITEM_WAVES 4         // items (waves) per workgroup of the item launch
#endif
#ifndef VH_ITEM_OCC
#define VH_ITEM_OCC 6           // waves per SIMD the item launch is compiled for
#endif
constexpr int kItemWaves = VH_ITEM_WAVES;

template <int kPatch>
__global__ __launch_bounds__(64 * kItemWaves, VH_ITEM_OCC) void raycast_items_kernel(const FrameParams fp, const DevPtrs dp, const RaycastArgs ra)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const RaycastSplit &sp = ra.sp;
    const unsigned long long stamp0 = ra.stamps ? __builtin_amdgcn_s_memrealtime() : 0ull;
    unsigned long long stampS = 0ull, stampW = 0ull;       // diagnostics: the queues are known; time spent in entered blocks
    int nDone = 0, nEntered = 0;
    const uint4 cn = reinterpret_cast<const uint4 *>(sp.counts)[lane];
    const uint32_t mine = cn.x + cn.y + cn.z + cn.w;
    uint32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(incl, d);
        incl += lane >= d ? t : 0u;
    }
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    const uint32_t excl = incl - mine;
    const uint32_t nWaves = gridDim.x * kItemWaves;
    const int prio[3] = {2, 0, 1};
    if (ra.stamps) stampS = __builtin_amdgcn_s_memrealtime();
    for (uint32_t it = blockIdx.x * kItemWaves + wave; it < total; it += nWaves) {
        ++nDone;
        // which queue: the first lane whose inclusive count exceeds the item's index, then one of its four
        const int l = __ffsll((long long)__ballot(it < incl)) - 1;
        uint32_t off = it - (uint32_t)__builtin_amdgcn_readlane((int)excl, l);
        const uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)cn.x, l), c1 = (uint32_t)__builtin_amdgcn_readlane((int)cn.y, l),
                       c2 = (uint32_t)__builtin_amdgcn_readlane((int)cn.z, l);
        uint32_t seg = (uint32_t)l * 4u;
        if (off >= c0) { off -= c0; ++seg; if (off >= c1) { off -= c1; ++seg; if (off >= c2) { off -= c2; ++seg; } } }
        const uint32_t itemIndex = seg * sp.segCap + off;
        const RcItem item = sp.items[itemIndex];
        const int kk[3] = {__builtin_amdgcn_readfirstlane(item.kx), __builtin_amdgcn_readfirstlane(item.ky), __builtin_amdgcn_readfirstlane(item.kz)};
        const int bptr = __builtin_amdgcn_readfirstlane(item.ptr);
        const uint32_t patch = (uint32_t)__builtin_amdgcn_readfirstlane((int)item.patch);
        const int py = (int)(patch / sp.patchesX), px = (int)(patch - (uint32_t)py * sp.patchesX);
        const int u = px * (kPatch == 0 ? 16 : 8) + (kPatch == 0 ? (lane & 15) : (lane & 7));
        const int v = py * (kPatch == 0 ? 4 : 8) + (kPatch == 0 ? (lane >> 4) : (lane >> 3));
        const bool inImage = u < fp.width && v < fp.height;
        // the ray's set-up, as the first launch left it
        const float *st = sp.state + (size_t)patch * (9 * 64) + lane;
        DdaAxis ax[3];
        int c[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            ax[a].G = ra.G[a];
            ax[a].E = st[64 * a];
            ax[a].invE = st[64 * (3 + a)];
            c[a] = __float_as_int(st[64 * (6 + a)]);
            ax[a].s = ax[a].E > 0.0f ? 1 : -1;
            ax[a].Gs = ax[a].E > 0.0f ? ax[a].G - 1.0f : ax[a].G;
        }
        unsigned long long *word = sp.best + (size_t)patch * 64 + lane;
        const unsigned long long cur = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool none = cur == kRcNone;
        const float bestT = none ? __builtin_inff() : rc_key_time((uint32_t)(cur >> 32));
        const int bestP = none ? 3 : (int)((cur >> 30) & 3ull);
        const CoopEntry e = coop_entry(ax, c, kk, ra.tMax);
        const float tE = e.tE;
        const int pE = e.pE, xe = e.xe;
        const bool inside = e.inside;
        const bool enters = inImage && e.enters && dda_before(tE, pE, bestT, bestP);      // (not behind the candidate the ray holds)
        const unsigned long long stampE = ra.stamps ? __builtin_amdgcn_s_memrealtime() : 0ull;
        if (__ballot(enters) != 0ull) ++nEntered;
        if (enters) {      // (structured: the lanes meet again behind the block, before the next item's wave-level operations)
            const float f0 = (float)ax[0].s, f1 = (float)ax[1].s, f2 = (float)ax[2].s;
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
            int candP = 0, candLin = 0;
            const Voxel *blk = dp.blocks + (size_t)bptr;
            while (walking) {
                const int lin = (int)((uint32_t)pl >> 16);
                const float vt = tArr;
                const int vp = pArr;
                const Voxel vv = blk[lin];
                // the crossing that ends this voxel (raycastSDF.frag:156-170)
                const bool m0 = tn0 < tn1 && tn0 < tn2;
                const bool m2 = !m0 && tn2 < tn1;
                const bool m1 = !m0 && !m2;
                tArr = m0 ? tn0 : m2 ? tn2 : tn1;
                pArr = m0 ? 2 : m2 ? 1 : 0;
                pl += m0 ? d0 : m2 ? d2 : d1;
                fc0 += m0 ? f0 : 0.0f; fc1 += m1 ? f1 : 0.0f; fc2 += m2 ? f2 : 0.0f;
                tn0 = (fc0 - gs0) * ie0; tn1 = (fc1 - gs1) * ie1; tn2 = (fc2 - gs2) * ie2;
                walking = tArr < ra.tMax && (pl & 0x6318) == 0x2108;     // (a voxel is visited iff the ray arrives before t_max)
                const bool valid = vv.weight > 0.0f;
                if (valid && vv.sdf <= 0.0f) {
                    if (firstVoxel) {
                        // the voxel the ray was in before the entry event: one step back on the entry axis, in the neighbouring block
                        const int vx = bx0 + (lin & 7), vy = by0 + ((lin >> 3) & 7), vz = bz0 + (lin >> 6);
                        const int n0 = vx - (xe == 0 ? ax[0].s : 0), n1 = vy - (xe == 1 ? ax[1].s : 0), n2 = vz - (xe == 2 ? ax[2].s : 0);
                        const int np = lookup_block(fp, dp, n0 >> 3, n1 >> 3, n2 >> 3);
                        pvd = false;
                        if (np != VH_FREE_BLOCK) {
                            const Voxel nb = dp.blocks[(size_t)np + (size_t)(((n2 & 7) << 6) | ((n1 & 7) << 3) | (n0 & 7))];
                            pvd = nb.weight > 0.0f; ps = nb.sdf;
                        }
                    }
                    if (pvd && ps > 0.0f) {                    // the block's first pair: nothing earlier behind it
                        have = dda_before(vt, vp, bestT, bestP);
                        candT = vt; candP = vp; candLin = lin;
                        walking = false;
                    }
                }
                pvd = valid; ps = vv.sdf;
                firstVoxel = false;
            }
            if (have) {
                const unsigned long long cand = ((unsigned long long)rc_time_key(candT) << 32) | ((unsigned long long)candP << 30) |
                                                ((unsigned long long)itemIndex << 9) | (unsigned long long)candLin;
                (void)__hip_atomic_fetch_min(word, cand, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (ra.stamps) stampW += __builtin_amdgcn_s_memrealtime() - stampE;
    }
    if (ra.stamps && lane == 0) {
        unsigned long long *o = ra.stamps + ((size_t)sp.patchesX * 0 + (size_t)ra.stampsItemBase + (size_t)(blockIdx.x * kItemWaves + wave)) * 8;
        o[0] = stamp0; o[1] = __builtin_amdgcn_s_memrealtime(); o[2] = (unsigned long long)nDone | ((unsigned long long)nEntered << 32);
        o[3] = stampS - stamp0; o[4] = stampW; o[5] = total; o[6] = 0ull; o[7] = 0ull;
    }
}

// ---------------------------------------------------------------------------
// The split form, third launch: the rays' words become pixels
// ---------------------------------------------------------------------------
// A word names the hit voxel (item -> block, voxel) and the event the ray arrived in it by; the pair's first sample is the
// voxel one step back on that event's axis (consecutive visited voxels).  Depth and normal with the fused kernel's
// arithmetic, in its order.  Workgroup 0 also clears the queues' counters for the next call.
template <int kPatch, bool kNormals>
__global__ __launch_bounds__(64 * kDdaBlockWaves) void raycast_resolve_kernel(const FrameParams fp, const DevPtrs dp, const RaycastArgs ra,
                                                                             float *__restrict__ depthOut, float4 *__restrict__ normalOut)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const RaycastSplit &sp = ra.sp;
    if (blockIdx.x == 0 && blockIdx.y == 0)
        for (int i = threadIdx.x; i < kRcSegs; i += 64 * kDdaBlockWaves) sp.counts[i] = 0u;
    int tx = blockIdx.x, ty = blockIdx.y;
    if (ra.xcdAware) {
        const int n = gridDim.x * gridDim.y, b = blockIdx.y * gridDim.x + blockIdx.x;
        if ((n & 7) == 0) {
            const int r = (b & 7) * (n >> 3) + (b >> 3);
            ty = r / (int)gridDim.x;
            tx = r - ty * (int)gridDim.x;
        }
    }
    const int pu = tx * 16 + (kPatch == 0 ? 0 : (wave & 1) * 8);
    const int pv = ty * 16 + (kPatch == 0 ? wave * 4 : (wave >> 1) * 8);
    const int u = pu + (kPatch == 0 ? (lane & 15) : (lane & 7));
    const int v = pv + (kPatch == 0 ? (lane >> 4) : (lane >> 3));
    if (u >= fp.width || v >= fp.height) return;
    const uint32_t patch = (uint32_t)(pv / (kPatch == 0 ? 4 : 8)) * sp.patchesX + (uint32_t)(pu / (kPatch == 0 ? 16 : 8));
    const unsigned long long w = sp.best[(size_t)patch * 64 + lane];
    if ((uint32_t)(w >> 32) == kRcDoneKey) return;                 // written by the first launch (per-lane walk)
    float hit = 0.0f;
    float4 n = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (w != kRcNone) {
        const RcItem item = sp.items[(uint32_t)(w >> 9) & 0x1fffffu];
        const int lin = (int)(w & 511ull), p = (int)((w >> 30) & 3ull);
        const int axis = p == 2 ? 0 : p == 0 ? 1 : 2;
        const int s = sp.state[(size_t)patch * (9 * 64) + 64 * axis + lane] > 0.0f ? 1 : -1;
        const int vx = (item.kx << 3) + (lin & 7), vy = (item.ky << 3) + ((lin >> 3) & 7), vz = (item.kz << 3) + (lin >> 6);
        const int p0 = vx - (axis == 0 ? s : 0), p1 = vy - (axis == 1 ? s : 0), p2 = vz - (axis == 2 ? s : 0);
        const float sdf = dp.blocks[(size_t)item.ptr + (size_t)lin].sdf;
        int pptr = item.ptr;
        if ((p0 >> 3) != item.kx || (p1 >> 3) != item.ky || (p2 >> 3) != item.kz) pptr = lookup_block(fp, dp, p0 >> 3, p1 >> 3, p2 >> 3);
        const float ps = dp.blocks[(size_t)pptr + (size_t)(((p2 & 7) << 6) | ((p1 & 7) << 3) | (p0 & 7))].sdf;
        // the samples sit at their voxels' centres: camera depth = row 2 of the inverse pose
        const float tc = ((ra.zrow[0] * (float)vx + ra.zrow[1] * (float)vy) + ra.zrow[2] * (float)vz) + ra.zrow[3];
        const float tp = ((ra.zrow[0] * (float)p0 + ra.zrow[1] * (float)p1) + ra.zrow[2] * (float)p2) + ra.zrow[3];
        hit = tp + ((tc - tp) * ps) / (ps - sdf);
        if (kNormals) n = dda_normal(fp, dp, vx, vy, vz, item.ptr);
    }
    depthOut[(size_t)v * fp.width + u] = hit;
    if (kNormals) normalOut[(size_t)v * fp.width + u] = n;
}

}  // namespace vh
