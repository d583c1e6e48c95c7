// vh_blocks.hip -- block silhouettes: the one render pass of the reference that works.
// SDFRenderer::drawToFrontAndBack (SDFRenderer.cpp:165-208) + depthWrite.{vert,geom,frag} rasterise
// one axis-aligned cube per compact entry (a unit cube in block units, scaled by voxelSize*8 through
// the MVP, Application.cpp:130-132) and keep the nearest front face per pixel (GL_LESS, :175); the
// design note (notes.md:3-16) wants the farthest back face as a second layer.  Here: per pixel the
// camera depth at which its ray enters the nearest / leaves the farthest cube of any allocated
// block, by an exact ray/box test (no rasteriser, no sampling): SURVEY.md 8(a) row R1.
// Oracle: vho_render_blocks.  Part of libvoxelhash_hip.so (gfx950); included by vh_kernels.hip.
#pragma once

namespace vh {

struct BlockView {
    float T[12];           // camera -> world of the view, rows 0..2
    float Tinv[12];        // world -> camera
    float fx, fy, cx, cy;
    float tMin, tMax;
};

constexpr uint32_t kFrontInit = 0x7f800000u;      // +inf: no cube in front of this pixel yet

// Two launches.  (1) blocks_list_kernel: every allocated entry (found through the bucket-occupancy bitmap)
// gets a 32-byte record {cube, screen bounding box of its corners} unless no pixel can see it.
// (2) blocks_tile_kernel: one wave per 8x8 pixel tile (a workgroup = a 16x16 region).  The workgroup first
// runs through the screen boxes alone (8 bytes per block, one per lane and pass) and lists the blocks whose
// box meets its 16x16 region (wave __ballot, one LDS atomic per wave and pass); the listed records are
// staged in LDS, each wave tests their boxes against its tile (one record per lane, __ballot) and every
// lane ray/box-tests its pixel against the records that overlap: nearest entry / farthest exit kept in
// registers, written once.  (Until round 6 every wave ran through ALL records -- 37 passes of 64 on C2's
// 2 366 blocks -- and every workgroup staged all 75 KB of them; listing first made the launch 5 - 15 %
// shorter, no more: what bounds it is the ray/box tests of the boxes that do meet a tile, ~60
// instructions each for 64 pixels: 35.9 us on that model, 13 us on a 120-frame one.)  No atomics, no image
// initialisation pass, no per-block load imbalance (round 1-2: one workgroup pass per block with
// atomicMin/atomicMax per covered pixel, 232 us on C2; this form: DESIGN.md 5).  The min / max over a
// pixel's cubes does not depend on the order, and a cube whose box misses a pixel fails that pixel's
// ray/box test anyway, so the images equal the oracle's bit for bit.
struct alignas(16) BlockRecord {
    float lo[3];             // the cube (block_cube)
    uint32_t xy0;            // x | y << 16: first pixel of the bounding box
    float hi[3];
    uint32_t xy1;            // last pixel; bit 31: every lo - o, hi - o is inside the fast-division range
};
static_assert(sizeof(BlockRecord) == 32, "BlockRecord");
constexpr uint32_t kBlockFastBit = 0x80000000u;    // (divisions: div_fixed, vh_device.h)

// the cube of block k: world [8k*vs, (8k+8)*vs] per axis (block2World of the min corner, no half-voxel shift)
__device__ __forceinline__ void block_cube(const FrameParams &fp, const int32_t pos[3], float lo[3], float hi[3])
{
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        lo[a] = (float)(int)((uint32_t)pos[a] * 8u) * fp.voxelSize;
        hi[a] = ((float)(int)((uint32_t)pos[a] * 8u) + 8.0f) * fp.voxelSize;
    }
}

// screen bounding box of what a pixel can see of the cube; false when no pixel can see it in [tMin, tMax].
// A ray meets the cube at camera depths [tNear, tFar] and is kept when that interval reaches into [tMin, tMax], so only the
// part of the cube at camera z >= tMin counts: a cube with corners nearer than that (or behind the camera) is bounded by its
// corners beyond the plane z = 0.999 tMin plus the points where its edges cross that plane -- the clipped cube is convex and
// wholly in front of the camera, so its projection is the hull of those points.  (Until round 6 such a cube took the whole
// image as its box: 31 - 57 of C2's ~450 visible blocks per view, and 88 % of all (block, tile) pairs the tile kernel tested.)
// With tMin < 0.05 a cube that reaches z <= 0.05 still takes the whole image.
__device__ __forceinline__ bool block_bounds(const FrameParams &fp, const BlockView &bv, const int32_t pos[3], int &x0,
                                             int &y0, int &x1, int &y1)
{
    float lo[3], hi[3];
    block_cube(fp, pos, lo, hi);
    float X[8], Y[8], Z[8];
    float zmin = 3.0e38f, zmax = -3.0e38f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float wx = (c & 1) ? hi[0] : lo[0], wy = (c & 2) ? hi[1] : lo[1], wz = (c & 4) ? hi[2] : lo[2];
        X[c] = bv.Tinv[0] * wx + bv.Tinv[1] * wy + bv.Tinv[2] * wz + bv.Tinv[3];
        Y[c] = bv.Tinv[4] * wx + bv.Tinv[5] * wy + bv.Tinv[6] * wz + bv.Tinv[7];
        Z[c] = bv.Tinv[8] * wx + bv.Tinv[9] * wy + bv.Tinv[10] * wz + bv.Tinv[11];
        zmin = __builtin_fminf(zmin, Z[c]);
        zmax = __builtin_fmaxf(zmax, Z[c]);
    }
    // wholly nearer than the first sample depth (e.g. behind the camera) or beyond the last: no pixel
    // can see it in [tMin, tMax] (camera depth along a ray = z of the point)
    if (zmax < bv.tMin - fp.voxelSize || zmin > bv.tMax + fp.voxelSize) return false;
    float umin = 3.0e38f, umax = -3.0e38f, vmin = 3.0e38f, vmax = -3.0e38f;
    auto take = [&](const float x, const float y, const float z) {
        const float iz = 1.0f / __builtin_fmaxf(z, 1.0e-6f);
        const float u = bv.fx * x * iz + bv.cx, v = bv.fy * y * iz + bv.cy;
        umin = __builtin_fminf(umin, u); umax = __builtin_fmaxf(umax, u);
        vmin = __builtin_fminf(vmin, v); vmax = __builtin_fmaxf(vmax, v);
    };
    const float zn = 0.999f * bv.tMin;
    x0 = 0; x1 = fp.width - 1; y0 = 0; y1 = fp.height - 1;
    if (zmin >= zn || zn < 0.05f) {
        if (zmin <= 0.05f) return true;                    // (whole image)
#pragma unroll
        for (int c = 0; c < 8; ++c) take(X[c], Y[c], Z[c]);
    } else {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (Z[c] >= zn) take(X[c], Y[c], Z[c]);
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                if (c & (1 << a)) continue;
                const int e = c | (1 << a);                // the edge from corner c along axis a
                if ((Z[c] >= zn) == (Z[e] >= zn)) continue;
                const float s = (zn - Z[c]) / (Z[e] - Z[c]);
                take(X[c] + s * (X[e] - X[c]), Y[c] + s * (Y[e] - Y[c]), zn);
            }
        }
        if (umin > umax) return false;                     // (nothing beyond the plane after all)
    }
    if (umax < -2.0f || vmax < -2.0f || umin > (float)fp.width + 1.0f || vmin > (float)fp.height + 1.0f) return false;
    x0 = max(0, (int)__builtin_floorf(__builtin_fmaxf(umin, -1.0e6f)) - 2);
    y0 = max(0, (int)__builtin_floorf(__builtin_fmaxf(vmin, -1.0e6f)) - 2);
    x1 = min(fp.width - 1, (int)__builtin_ceilf(__builtin_fminf(umax, 1.0e6f)) + 2);
    y1 = min(fp.height - 1, (int)__builtin_ceilf(__builtin_fminf(vmax, 1.0e6f)) + 2);
    return x0 <= x1 && y0 <= y1;
}

// counts: two words used in turn by successive calls (call n appends through counts[n & 1] and zeroes the
// other for call n + 1), so no launch is spent on a reset
__global__ __launch_bounds__(256) void blocks_list_kernel(const FrameParams fp, const DevPtrs dp, const BlockView bv,
                                                          BlockRecord *__restrict__ records, uint2 *__restrict__ bounds,
                                                          int32_t capacity, int32_t *__restrict__ counts, int parity)
{
    const uint32_t owned = fp.bucketHi - fp.bucketLo;
    const uint32_t w = blockIdx.x * 256 + threadIdx.x;
    if (w == 0) counts[parity ^ 1] = 0;
    if (w >= (owned + 31u) / 32u) return;
    uint32_t bits = dp.bucketBits[w];
    const bool holes = (fp.flags & kFlagOverflow) != 0u;        // entries form a prefix of the bucket unless chains leave holes
    while (bits != 0u) {
        const uint32_t bucket = w * 32u + (uint32_t)__ffs((int)bits) - 1u;
        bits &= bits - 1u;
        for (uint32_t s = 0; s < fp.bucketSize; ++s) {
            const VoxelEntry e = dp.table[(size_t)bucket * fp.bucketSize + s];
            if (e.ptr == VH_FREE_BLOCK) {
                if (holes) continue;
                break;
            }
            BlockRecord r;
            int x0, y0, x1, y1;
            if (!block_bounds(fp, bv, e.pos, x0, y0, x1, y1)) continue;
            block_cube(fp, e.pos, r.lo, r.hi);
            r.xy0 = (uint32_t)x0 | ((uint32_t)y0 << 16);
            r.xy1 = (uint32_t)x1 | ((uint32_t)y1 << 16);
            bool fast = true;
#pragma unroll
            for (int a = 0; a < 3; ++a)
                fast = fast && fast_range(r.lo[a] - bv.T[4 * a + 3], 0x1p-50f, 0x1p50f) &&
                       fast_range(r.hi[a] - bv.T[4 * a + 3], 0x1p-50f, 0x1p50f);
            if (fast) r.xy1 |= kBlockFastBit;
            const int slot = atomicAdd(counts + parity, 1);
            if (slot < capacity) {
                records[slot] = r;
                bounds[slot] = make_uint2(r.xy0, r.xy1);      // the screen boxes alone: what a workgroup's first pass reads
            }
        }
    }
}

// same operations in the same order as the oracle's ray_box
__device__ __forceinline__ bool ray_box(const float o[3], const float d[3], const float lo[3], const float hi[3],
                                        float &tNear, float &tFar)
{
    tNear = -3.0e38f;
    tFar = 3.0e38f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (d[a] == 0.0f) {
            if (o[a] < lo[a] || o[a] > hi[a]) return false;
            continue;
        }
        const float t0 = (lo[a] - o[a]) / d[a], t1 = (hi[a] - o[a]) / d[a];
        tNear = __builtin_fmaxf(tNear, __builtin_fminf(t0, t1));
        tFar = __builtin_fminf(tFar, __builtin_fmaxf(t0, t1));
    }
    return tNear <= tFar;
}

constexpr int kBlocksStage = 256;        // records staged per round (one per lane of the workgroup)
constexpr int kBlocksChunk = 4 * kBlocksStage;     // boxes a workgroup filters before it works its list off (the list cannot overflow)

__global__ __launch_bounds__(256) void blocks_tile_kernel(const FrameParams fp, const BlockView bv,
                                                          const BlockRecord *__restrict__ records,
                                                          const uint2 *__restrict__ bounds, int32_t capacity,
                                                          const int32_t *__restrict__ counts, int parity,
                                                          float *__restrict__ front, float *__restrict__ back)
{
    __shared__ BlockRecord stage[kBlocksStage];
    __shared__ uint32_t list[kBlocksChunk];
    __shared__ int listCount;
    const int n = min(counts[parity], capacity);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & (kWave - 1);
    // the workgroup's 16x16 region and, inside it, the wave's 8x8 tile; lane -> pixel row-major in the tile
    const int rx0 = (int)blockIdx.x * 16, ry0 = (int)blockIdx.y * 16;
    const int tx0 = rx0 + (wave & 1) * 8, ty0 = ry0 + (wave >> 1) * 8;
    const int px = tx0 + (lane & 7), py = ty0 + (lane >> 3);
    const float o[3] = {bv.T[3], bv.T[7], bv.T[11]};
    const float dx = ((float)px - bv.cx) / bv.fx, dy = ((float)py - bv.cy) / bv.fy;
    const float d[3] = {bv.T[0] * dx + bv.T[1] * dy + bv.T[2], bv.T[4] * dx + bv.T[5] * dy + bv.T[6],
                        bv.T[8] * dx + bv.T[9] * dy + bv.T[10]};
    uint32_t nearest = kFrontInit, farthest = 0u;               // positive floats order like their bit patterns
    // per pixel and axis: a zero direction component (the slab test becomes an inside test, as in ray_box),
    // the refined reciprocal, and whether all three divisors are inside the fast-division range
    bool zero[3], fastPixel = true;
    float r1[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        zero[a] = d[a] == 0.0f;
        r1[a] = refined_rcp(d[a]);
        fastPixel = fastPixel && (zero[a] || fast_range(d[a], 0x1p-40f, 0x1p40f));
    }
    const bool fastWave = __ballot(!fastPixel) == 0ull;
    for (int chunk = 0; chunk < n; chunk += kBlocksChunk) {
        // ---- the boxes of this chunk against the workgroup's region: the ones that meet it, listed in LDS
        uint2 box[kBlocksChunk / kBlocksStage];
#pragma unroll
        for (int p = 0; p < kBlocksChunk / kBlocksStage; ++p) {
            const int idx = chunk + p * kBlocksStage + (int)threadIdx.x;
            box[p] = idx < n ? bounds[idx] : make_uint2(0xffffffffu, 0u);       // (an empty box: x0 = 65535 > x1 = 0)
        }
        if (threadIdx.x == 0) listCount = 0;
        __syncthreads();                                                        // (also: the last round's readers of stage / list are done)
#pragma unroll
        for (int p = 0; p < kBlocksChunk / kBlocksStage; ++p) {
            const int x0 = (int)(box[p].x & 0xffffu), y0 = (int)(box[p].x >> 16), x1 = (int)(box[p].y & 0xffffu),
                      y1 = (int)((box[p].y >> 16) & 0x7fffu);
            const bool meets = x0 <= rx0 + 15 && x1 >= rx0 && y0 <= ry0 + 15 && y1 >= ry0;
            const unsigned long long m = __ballot(meets);
            if (m != 0ull) {
                int at = 0;
                if (lane == 0) at = atomicAdd(&listCount, __popcll(m));
                at = __shfl(at, 0);
                if (meets) list[at + __popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)(chunk + p * kBlocksStage + (int)threadIdx.x);
            }
        }
        __syncthreads();
        const int listed = listCount;
        // ---- the listed records, 256 at a time through LDS (the order of the list is whatever the waves' atomics made it:
        // a minimum and a maximum do not depend on it)
        for (int base = 0; base < listed; base += kBlocksStage) {
            if (base > 0) __syncthreads();
            if (base + (int)threadIdx.x < listed) stage[threadIdx.x] = records[list[base + threadIdx.x]];
            __syncthreads();
            const int count = min(kBlocksStage, listed - base);
            for (int k = 0; k < count; k += kWave) {
                bool overlap = false;
                if (k + lane < count) {
                    const uint32_t xy0 = stage[k + lane].xy0, xy1 = stage[k + lane].xy1;
                    const int x0 = (int)(xy0 & 0xffffu), y0 = (int)(xy0 >> 16), x1 = (int)(xy1 & 0xffffu), y1 = (int)((xy1 >> 16) & 0x7fffu);
                    overlap = x0 <= tx0 + 7 && x1 >= tx0 && y0 <= ty0 + 7 && y1 >= ty0;
                }
                unsigned long long mask = __ballot(overlap);
                while (mask != 0ull) {
                    const int src = __ffsll((long long)mask) - 1;
                    mask &= mask - 1ull;
                    const BlockRecord q = stage[k + src];            // one LDS address for the wave: broadcast read
                    float tNear, tFar;
                    bool hit;
                    if (fastWave && (q.xy1 & kBlockFastBit)) {
                        // ray_box without branches, divisions by the pixel's fixed direction
                        tNear = -3.0e38f;
                        tFar = 3.0e38f;
                        hit = true;
#pragma unroll
                        for (int a = 0; a < 3; ++a) {
                            const float t0 = div_fixed(q.lo[a] - o[a], d[a], r1[a]), t1 = div_fixed(q.hi[a] - o[a], d[a], r1[a]);
                            const float tn = zero[a] ? -3.0e38f : __builtin_fminf(t0, t1);
                            const float tf = zero[a] ? 3.0e38f : __builtin_fmaxf(t0, t1);
                            hit = hit && !(zero[a] && (o[a] < q.lo[a] || o[a] > q.hi[a]));
                            tNear = __builtin_fmaxf(tNear, tn);
                            tFar = __builtin_fminf(tFar, tf);
                        }
                        hit = hit && tNear <= tFar;
                    } else {
                        hit = ray_box(o, d, q.lo, q.hi, tNear, tFar);
                    }
                    if (!hit || tFar < bv.tMin || tNear > bv.tMax) continue;
                    // (+ 0.0f: -0 from a face through the camera centre becomes +0, as in the oracle)
                    nearest = min(nearest, __float_as_uint(__builtin_fmaxf(tNear, bv.tMin) + 0.0f));
                    farthest = max(farthest, __float_as_uint(__builtin_fminf(tFar, bv.tMax) + 0.0f));
                }
            }
        }
    }
    if (px < fp.width && py < fp.height) {
        const size_t idx = (size_t)py * fp.width + px;
        front[idx] = nearest == kFrontInit ? 0.0f : __uint_as_float(nearest);       // no cube: 0
        back[idx] = __uint_as_float(farthest);
    }
}

}  // namespace vh
