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
// Because the state of the walk is (integer voxel, tnext of three integer coordinates), leaving an empty cell
// in ONE step can be made exact: the exit event is the first of the cell's three boundary events in merge order,
// and every other axis advances past exactly those of its events that precede it (a float estimate of the
// coordinate, then the merge predicate itself decides, so the estimate's rounding never matters).  The image
// therefore does not depend on which cells are skipped, and the kernel may skip on hashed bitmaps whose stale or
// colliding bits only make it skip less: an empty 4x4x4-block macro cell (32 voxels) or an absent block (8).
// Instruction budget per ray on C2 (PMC, profiles/): a handful of macro-cell jumps, a few block look-ups, then
// ~10-20 voxel steps of ~30 VALU instructions inside the surface block -- against 40 fully re-derived samples
// (4x4 transform, three IEEE divisions, three floor divisions: ~100 instructions each) of the fixed-step march.
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

struct RaycastArgs {
    float fx, fy, cx, cy;
    float tMin, tMax;
    float zrow[4];            // row 2 of the inverse pose, first three scaled by voxelSize: camera depth of a voxel centre
    int budget;               // hang guard: more steps than any ray of this view can take (host: vh_raycast)
    int xcdAware;
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

template <int kPatch, bool kNormals>
__global__ __launch_bounds__(256) void raycast_dda_kernel(const FrameParams fp, const DevPtrs dp, const RaycastArgs ra,
                                                          float *__restrict__ depthOut, float4 *__restrict__ normalOut)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int tx = blockIdx.x, ty = blockIdx.y;
    if (ra.xcdAware) {                       // each XCD (own L2) renders a contiguous run of image tiles
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
    const float dx = ((float)u - ra.cx) / ra.fx;
    const float dy = ((float)v - ra.cy) / ra.fy;
    const float vs = fp.voxelSize;
    DdaAxis ax[3];
    int c[3];
    float tn[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float D = fp.T[4 * a + 0] * dx + fp.T[4 * a + 1] * dy + fp.T[4 * a + 2];
        ax[a].G = fp.T[4 * a + 3] / vs + 0.5f;
        ax[a].E = D / vs;
        const bool active = __builtin_fabsf(ax[a].E) > 1.0e-20f;
        ax[a].invE = active ? 1.0f / ax[a].E : 0.0f;
        ax[a].s = ax[a].E > 0.0f ? 1 : -1;
        ax[a].Gs = ax[a].E > 0.0f ? ax[a].G - 1.0f : ax[a].G;
        c[a] = f2i_rz(__builtin_floorf(ax[a].G + ax[a].E * ra.tMin));
        tn[a] = dda_tnext(ax[a], c[a]);
    }
    const int prio[3] = {2, 0, 1};
    bool prevValid = false, found = false, done = false;
    float prevSdf = 0.0f, prevT = 0.0f, hit = 0.0f;
    int hx = 0, hy = 0, hz = 0, hptr = VH_FREE_BLOCK;          // the hit voxel and its block (normals)
    bool haveMacro = false, macroEmpty = false;
    int cmx = 0, cmy = 0, cmz = 0;
    int budget = ra.budget;
    while (!done && budget > 0) {
        // ---- the cell the ray stands in: macro cell first, then the block ----
        const int kx = c[0] >> 3, ky = c[1] >> 3, kz = c[2] >> 3;          // voxel2Block for two's complement ints
        const int mx = kx >> 2, my = ky >> 2, mz = kz >> 2;
        if (!haveMacro || mx != cmx || my != cmy || mz != cmz) {
            cmx = mx; cmy = my; cmz = mz;
            haveMacro = true;
            const uint32_t hm = macro_hash(mx, my, mz);
            macroEmpty = !((dp.macroBits[hm >> 5] >> (hm & 31u)) & 1u);
        }
        const int cptr = macroEmpty ? VH_FREE_BLOCK : lookup_block(fp, dp, kx, ky, kz);
        if (cptr == VH_FREE_BLOCK) {
            // ---- leave the empty cell in one step ----
            --budget;
            prevValid = false;
            const int shift = macroEmpty ? 5 : 3, span = macroEmpty ? 31 : 7;
            int cs[3];
            float te[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                cs[a] = (int)((uint32_t)(c[a] >> shift) << shift) + (ax[a].s > 0 ? span : 0);   // last coordinate inside the cell
                te[a] = dda_tnext(ax[a], cs[a]);
            }
            const int x = (te[0] < te[1] && te[0] < te[2]) ? 0 : (te[2] < te[1]) ? 2 : 1;
            const float tex = x == 0 ? te[0] : x == 1 ? te[1] : te[2];
            if (!(tex < ra.tMax)) { done = true; break; }            // the ray ends inside the empty cell
            const int px = x == 0 ? 2 : x == 1 ? 0 : 1;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                c[a] = (a == x) ? cs[a] + ax[a].s : dda_advance(ax[a], prio[a], c[a], cs[a], tex, px);
                tn[a] = dda_tnext(ax[a], c[a]);
            }
            continue;
        }
        // ---- an allocated block: voxel by voxel until the ray leaves it ----
        for (;;) {
            --budget;
            const Voxel s = dp.blocks[(size_t)cptr + (size_t)(((c[2] & 7) << 6) | ((c[1] & 7) << 3) | (c[0] & 7))];
            const float tc = ((ra.zrow[0] * (float)c[0] + ra.zrow[1] * (float)c[1]) + ra.zrow[2] * (float)c[2]) + ra.zrow[3];
            const int vx = c[0], vy = c[1], vz = c[2];
            // the crossing that ends this voxel (raycastSDF.frag:156-170), taken before the sample is looked at:
            // the step does not depend on the voxel's contents, so it runs under the load
            const int a = (tn[0] < tn[1] && tn[0] < tn[2]) ? 0 : (tn[2] < tn[1]) ? 2 : 1;
            const float tOut = a == 0 ? tn[0] : a == 1 ? tn[1] : tn[2];
            bool left;
            if (a == 0) { c[0] += ax[0].s; tn[0] = dda_tnext(ax[0], c[0]); left = (c[0] >> 3) != kx; }
            else if (a == 1) { c[1] += ax[1].s; tn[1] = dda_tnext(ax[1], c[1]); left = (c[1] >> 3) != ky; }
            else { c[2] += ax[2].s; tn[2] = dda_tnext(ax[2], c[2]); left = (c[2] >> 3) != kz; }
            if (s.weight > 0.0f) {
                if (prevValid && prevSdf > 0.0f && s.sdf <= 0.0f) {
                    hit = prevT + ((tc - prevT) * prevSdf) / (prevSdf - s.sdf);
                    found = true; done = true;
                    hx = vx; hy = vy; hz = vz; hptr = cptr;
                    break;
                }
                prevValid = true; prevSdf = s.sdf; prevT = tc;
            } else {
                prevValid = false;
            }
            if (!(tOut < ra.tMax)) { done = true; break; }
            if (left || budget <= 0) break;
        }
    }
    depthOut[(size_t)v * fp.width + u] = hit;
    if (!kNormals) return;
    // ---- normal of the hit: TSDF gradient at the hit voxel, normalised, camera frame, w = 0 ----
    float4 n = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (found) {
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
    }
    normalOut[(size_t)v * fp.width + u] = n;
}

}  // namespace vh
