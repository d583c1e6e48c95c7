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

__global__ __launch_bounds__(256) void blocks_init_kernel(uint32_t *front, uint32_t *back, int32_t n, int32_t *listCount)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) *listCount = 0;
    if (i < n) { front[i] = kFrontInit; back[i] = 0u; }
}

// allocated entries, found through the bucket-occupancy bitmap (one lane per 32-bucket word)
__global__ __launch_bounds__(256) void blocks_list_kernel(const FrameParams fp, const DevPtrs dp, int32_t *list,
                                                          int32_t capacity, int32_t *listCount)
{
    const uint32_t owned = fp.bucketHi - fp.bucketLo;
    const uint32_t w = blockIdx.x * 256 + threadIdx.x;
    if (w >= (owned + 31u) / 32u) return;
    uint32_t bits = dp.bucketBits[w];
    while (bits != 0u) {
        const uint32_t bucket = w * 32u + (uint32_t)__ffs((int)bits) - 1u;
        bits &= bits - 1u;
        for (uint32_t s = 0; s < fp.bucketSize; ++s) {
            const uint32_t e = bucket * fp.bucketSize + s;
            if (dp.table[e].ptr == VH_FREE_BLOCK) break;               // entries form a prefix
            const int slot = atomicAdd(listCount, 1);
            if (slot < capacity) list[slot] = (int32_t)e;
        }
    }
}

// the cube of block k: world [8k*vs, (8k+8)*vs] per axis (block2World of the min corner, no half-voxel shift)
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

// one listed block per workgroup pass: screen bounding box of the cube's corners (the whole image when
// a corner is at or behind the camera plane), every pixel in it tested exactly
__global__ __launch_bounds__(256) void blocks_raster_kernel(const FrameParams fp, const DevPtrs dp, const BlockView bv,
                                                            const int32_t *__restrict__ list, int32_t capacity,
                                                            const int32_t *__restrict__ listCount,
                                                            uint32_t *__restrict__ front, uint32_t *__restrict__ back)
{
    const int n = min(*listCount, capacity);
    for (int b = blockIdx.x; b < n; b += gridDim.x) {
        const VoxelEntry e = dp.table[list[b]];
        float lo[3], hi[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            lo[a] = (float)(int)((uint32_t)e.pos[a] * 8u) * fp.voxelSize;
            hi[a] = ((float)(int)((uint32_t)e.pos[a] * 8u) + 8.0f) * fp.voxelSize;
        }
        float zmin = 3.0e38f, zmax = -3.0e38f, umin = 3.0e38f, umax = -3.0e38f, vmin = 3.0e38f, vmax = -3.0e38f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float wx = (c & 1) ? hi[0] : lo[0], wy = (c & 2) ? hi[1] : lo[1], wz = (c & 4) ? hi[2] : lo[2];
            const float x = bv.Tinv[0] * wx + bv.Tinv[1] * wy + bv.Tinv[2] * wz + bv.Tinv[3];
            const float y = bv.Tinv[4] * wx + bv.Tinv[5] * wy + bv.Tinv[6] * wz + bv.Tinv[7];
            const float z = bv.Tinv[8] * wx + bv.Tinv[9] * wy + bv.Tinv[10] * wz + bv.Tinv[11];
            zmin = __builtin_fminf(zmin, z);
            zmax = __builtin_fmaxf(zmax, z);
            const float iz = 1.0f / __builtin_fmaxf(z, 1.0e-6f);
            const float u = bv.fx * x * iz + bv.cx, v = bv.fy * y * iz + bv.cy;
            umin = __builtin_fminf(umin, u); umax = __builtin_fmaxf(umax, u);
            vmin = __builtin_fminf(vmin, v); vmax = __builtin_fmaxf(vmax, v);
        }
        // wholly nearer than the first sample depth (e.g. behind the camera) or beyond the last: no pixel
        // can see it in [tMin, tMax] (camera depth along a ray = z of the point)
        if (zmax < bv.tMin - fp.voxelSize || zmin > bv.tMax + fp.voxelSize) continue;
        int x0 = 0, x1 = fp.width - 1, y0 = 0, y1 = fp.height - 1;
        if (zmin > 0.05f) {
            if (umax < -2.0f || vmax < -2.0f || umin > (float)fp.width + 1.0f || vmin > (float)fp.height + 1.0f) continue;
            x0 = max(0, (int)__builtin_floorf(umin) - 2);
            y0 = max(0, (int)__builtin_floorf(vmin) - 2);
            x1 = min(fp.width - 1, (int)__builtin_ceilf(__builtin_fminf(umax, 1.0e6f)) + 2);
            y1 = min(fp.height - 1, (int)__builtin_ceilf(__builtin_fminf(vmax, 1.0e6f)) + 2);
        }
        // blockIdx.y cuts the box into horizontal bands (a block next to the camera covers 10^5 pixels)
        const int bandH = (y1 - y0 + (int)gridDim.y) / (int)gridDim.y;
        y0 += (int)blockIdx.y * bandH;
        y1 = min(y1, y0 + bandH - 1);
        if (y0 > y1) continue;
        const int bw = x1 - x0 + 1, bh = y1 - y0 + 1;
        const float o[3] = {bv.T[3], bv.T[7], bv.T[11]};
        for (int i = threadIdx.x; i < bw * bh; i += 256) {
            const int py = y0 + i / bw, px = x0 + (i - (i / bw) * bw);
            const float dx = ((float)px - bv.cx) / bv.fx, dy = ((float)py - bv.cy) / bv.fy;
            const float d[3] = {bv.T[0] * dx + bv.T[1] * dy + bv.T[2], bv.T[4] * dx + bv.T[5] * dy + bv.T[6],
                                bv.T[8] * dx + bv.T[9] * dy + bv.T[10]};
            float tNear, tFar;
            if (!ray_box(o, d, lo, hi, tNear, tFar)) continue;
            if (tFar < bv.tMin || tNear > bv.tMax) continue;
            const float f = __builtin_fmaxf(tNear, bv.tMin), k = __builtin_fminf(tFar, bv.tMax);
            atomicMin(front + (size_t)py * fp.width + px, __float_as_uint(f));      // positive floats order like uints
            atomicMax(back + (size_t)py * fp.width + px, __float_as_uint(k));
        }
    }
}

// +inf (no cube) -> 0 in the front layer
__global__ __launch_bounds__(256) void blocks_finish_kernel(uint32_t *front, int32_t n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n && front[i] == kFrontInit) front[i] = 0u;
}

}  // namespace vh
