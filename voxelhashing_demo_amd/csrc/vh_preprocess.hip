// vh_preprocess.hip -- depth pre-processing, table set-up kernels, device-side test hook.
// Part of libvoxelhash_hip.so (gfx950); included by vh_kernels.hip after vh_device.h.
#pragma once

namespace vh {

// ---------------------------------------------------------------------------
// depth pre-processing (preProcess, CameraTrackingUtils.cu:115-120)
// ---------------------------------------------------------------------------
// calculateVertexPositions (:50-73) and calculateNormals (:75-113) as ONE kernel: the
// reference writes the vertex map, synchronises, and reads it back five times per pixel for
// the normals; here the four neighbour vertices are recomputed from the 2-byte depth (same
// arithmetic, same bits), so the pass reads 2 B and writes 32 B per pixel.
struct Mat3 { float m[9]; };

__device__ __forceinline__ float3 vertex_from_depth(const uint16_t *__restrict__ depth, const Mat3 &kinv, int W,
                                                    int x, int y)
{
    const float d = (float)depth[(size_t)y * W + x] / 5000.0f;          // :64, 5000 units = 1 m
    const float fx = (float)x, fy = (float)y;
    const float px = kinv.m[0] * fx + kinv.m[1] * fy + kinv.m[2] * 1.0f; // K_inv * (x, y, 1)  :71
    const float py = kinv.m[3] * fx + kinv.m[4] * fy + kinv.m[5] * 1.0f;
    const float pz = kinv.m[6] * fx + kinv.m[7] * fy + kinv.m[8] * 1.0f;
    return make_float3(px * d, py * d, pz * d);
}

__global__ __launch_bounds__(256) void preprocess_kernel(const uint16_t *__restrict__ depth, const Mat3 kinv, int W,
                                                         int H, float4 *__restrict__ positions,
                                                         float4 *__restrict__ normals)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= W * H) return;
    const int y = idx / W, x = idx - y * W;
    const float3 cc = vertex_from_depth(depth, kinv, W, x, y);
    positions[idx] = make_float4(cc.x, cc.y, cc.z, 1.0f);                                    // :73
    float4 n = make_float4(0.0f, 0.0f, 0.0f, 0.0f);                                          // :90
    if (x > 0 && x < W - 1 && y > 0 && y < H - 1) {                                          // :92
        const float3 pc = vertex_from_depth(depth, kinv, W, x, y + 1);
        const float3 cp = vertex_from_depth(depth, kinv, W, x + 1, y);
        const float3 mc = vertex_from_depth(depth, kinv, W, x, y - 1);
        const float3 cm = vertex_from_depth(depth, kinv, W, x - 1, y);
        if (cc.x != 0.0f && pc.x != 0.0f && cp.x != 0.0f && mc.x != 0.0f && cm.x != 0.0f) { // :100
            const float ax = pc.x - mc.x, ay = pc.y - mc.y, az = pc.z - mc.z;
            const float bx = cp.x - cm.x, by = cp.y - cm.y, bz = cp.z - cm.z;
            const float nx = ay * bz - az * by, ny = az * bx - ax * bz, nz = ax * by - ay * bx;   // cross
            const float l = __builtin_sqrtf(nx * nx + ny * ny + nz * nz);                          // length
            if (l > 0.0f) n = make_float4(nx / l, ny / l, nz / l, 0.0f);                     // :105-109
        }
    }
    normals[idx] = n;
}

// ---------------------------------------------------------------------------
// set-up kernels (deviceAllocate, VoxelUtils.cu:151-166)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void reset_table_kernel(VoxelEntry *table, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        VoxelEntry e;
        e.pos[0] = e.pos[1] = e.pos[2] = VH_POS_SENTINEL;
        e.ptr = VH_FREE_BLOCK;
        e.offset = 0;
        table[i] = e;
    }
}

__global__ __launch_bounds__(256) void reset_heap_kernel(uint32_t *heap, uint32_t n)
{
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) heap[i] = i;
}

// debug / known-answer hook: runs the scalar helpers on n points
__global__ void debug_eval_kernel(const FrameParams fp, const float4 *__restrict__ pts, int n, int32_t *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = pts[i];
    const int3_ b = world2block(p.x, p.y, p.z, fp.voxelSize);
    int sx, sy;
    project(fp.proj, p.x, p.y, p.z, sx, sy);
    int32_t *o = out + (size_t)i * 8;
    o[0] = b.x; o[1] = b.y; o[2] = b.z;
    o[3] = (int32_t)hash_block(b.x, b.y, b.z, fp.numBuckets);
    o[4] = block_in_frustum(fp, b.x, b.y, b.z) ? 1 : 0;
    o[5] = sx; o[6] = sy;
    o[7] = f2i_rz(p.w);
}

}  // namespace vh
