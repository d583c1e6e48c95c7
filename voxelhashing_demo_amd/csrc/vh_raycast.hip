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

// Workgroups are handed to the 8 XCDs round robin (workgroup b runs on XCD b % 8, each with its own L2).  The tiles are
// renumbered so that an XCD gets a contiguous run of image tiles: neighbouring rays, which sample the same blocks, share an L2.
__device__ __forceinline__ void xcd_tile(int &tx, int &ty)
{
    const int n = gridDim.x * gridDim.y, b = blockIdx.y * gridDim.x + blockIdx.x;
    if ((n & 7) == 0) {
        const int r = (b & 7) * (n >> 3) + (b >> 3);
        ty = r / (int)gridDim.x;
        tx = r - ty * (int)gridDim.x;
    }
}

// A wave is an 8x8-pixel patch of its workgroup's 16x16 tile (16x4 rows measured slower in round 2).
__global__ __launch_bounds__(256) void raycast_kernel(const FrameParams fp, const DevPtrs dp, float fx, float fy,
                                                      float cx, float cy, float tMin, int nSteps,
                                                      float *__restrict__ depthOut)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int tx = blockIdx.x, ty = blockIdx.y;
    xcd_tile(tx, ty);
    const int u = tx * 16 + (wave & 1) * 8 + (lane & 7);
    const int v = ty * 16 + (wave >> 1) * 8 + (lane >> 3);
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

constexpr int kDdaBlockWaves = 4;        // waves (8x8 pixel patches) per workgroup of the DDA kernels
// Waves per SIMD the register budget of the DDA kernels must allow.  A 640x480 view is 4 800 waves = 4.7 per SIMD: with 4
// resident the last 704 start when the first ones end; with 5 (96 registers, a few spilled) all start at once: 40.6 -> 35.3 us
// in round 3.  6 (80 registers, 30-60 spilled): slower.
#define VH_DDA_WAVES 5

struct RaycastArgs {
    float fx, fy, cx, cy;
    float tMin, tMax;
    float zrow[4];            // row 2 of the inverse pose, first three scaled by voxelSize: camera depth of a voxel centre
    float G[3];               // pose translation / voxelSize + 0.5: the camera centre in voxel-grid units (same for every ray)
    float invVs;              // 1 / voxelSize (beam boxes only)
    int budget;               // hang guard: more steps than any ray of this view can take (host: vh_raycast)
    int beam;                 // 2: the cooperative form (raycast_coop_kernel); 1: per-lane walk behind a beam front end; 0: per-lane walk
                              // from t_min (views with t_min <= 0)
    int patchesX, numPatches; // cooperative form: 8x8-pixel patches per image row / in the image
    int groups;               // ... and its workgroups: group g renders one patch of each quarter of the patch grid
    unsigned long long *stamps;   // diagnostics (tools/raycast_stamps.py): 8 words per wave, or null
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
// ray set-up, beam front end and per-lane walk (RaycastArgs::beam 1 / 0, and the cooperative form's fall-back)
// ---------------------------------------------------------------------------
__device__ __forceinline__ void dda_ray(const FrameParams &fp, const RaycastArgs &ra, int u, int v, DdaAxis (&ax)[3], int (&c)[3],
                                        float &dx, float &dy)
{
    const float vs = fp.voxelSize;
    dx = ((float)u - ra.cx) / ra.fx;
    dy = ((float)v - ra.cy) / ra.fy;
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
}

// Beam front end of a wave whose 64 lanes are an 8x8 pixel patch: the depth before which no ray of the wave can meet an
// allocated block; every live lane's walk starts there (c is advanced), or the lanes are dead (nothing allocated ahead).
__device__ __forceinline__ void dda_front_end(const FrameParams &fp, const DevPtrs &dp, const RaycastArgs &ra, float dx, float dy,
                                              const DdaAxis (&ax)[3], int (&c)[3], bool &live)
{
    const int lane = threadIdx.x & 63;
    const float vs = fp.voxelSize;
    Beam bm;
    // (the corner rays' directions are those of the patch's corner lanes: no division here)
    const float a0 = __shfl(dx, 0), a1 = __shfl(dx, 7);
    const float b0 = __shfl(dy, 0), b1 = __shfl(dy, 56);
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

struct DdaHit {
    float hit;                 // camera depth of the surface (0: none)
    bool found;
    int hx, hy, hz, hptr;      // the hit voxel and its block (normals)
    int steps, rounds;         // diagnostics
};

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
template <int kDdaK>      // cells enumerated ahead per round: 2 in the per-lane kernel, 1 where the walk is a rare fall-back (fewer registers)
__device__ __forceinline__ DdaHit dda_lane_walk(const FrameParams &fp, const DevPtrs &dp, const RaycastArgs &ra, const DdaAxis (&ax)[3],
                                                const int (&c)[3], bool live)
{
    const int prio[3] = {2, 0, 1};
    bool found = false;
    float hit = 0.0f;
    int hx = 0, hy = 0, hz = 0, hptr = VH_FREE_BLOCK;          // the last valid sample's voxel; after a hit: the hit voxel and its block
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
    return DdaHit{hit, found, hx, hy, hz, hptr, ra.budget - budget, round};
}

template <bool kNormals>
__global__ __launch_bounds__(64 * kDdaBlockWaves, VH_DDA_WAVES) void raycast_dda_kernel(const FrameParams fp, const DevPtrs dp, const RaycastArgs ra,
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
    bool live = inImage;
    if (ra.beam) dda_front_end(fp, dp, ra, dx, dy, ax, c, live);
    const unsigned long long stamp1 = ra.stamps ? __builtin_amdgcn_s_memrealtime() : 0ull;
    const DdaHit h = dda_lane_walk<2>(fp, dp, ra, ax, c, live);
    if (ra.stamps && lane == 0) {
        const size_t w = ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * kDdaBlockWaves + wave) * 8;
        ra.stamps[w] = stamp0; ra.stamps[w + 1] = __builtin_amdgcn_s_memrealtime();
        ra.stamps[w + 2] = (unsigned long long)h.steps | ((unsigned long long)h.rounds << 32);
        ra.stamps[w + 3] = (unsigned long long)(pu | (pv << 16)) | ((stamp1 - stamp0) << 32);
        ra.stamps[w + 4] = 0ull; ra.stamps[w + 5] = 0ull; ra.stamps[w + 6] = 0ull; ra.stamps[w + 7] = 0ull;
    }
    if (!inImage) return;
    depthOut[(size_t)v * fp.width + u] = h.hit;
    if (!kNormals) return;
    normalOut[(size_t)v * fp.width + u] = h.found ? dda_normal(fp, dp, h.hx, h.hy, h.hz, h.hptr) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

}  // namespace vh
