// vh_icp.hip -- frame-to-frame point-to-plane ICP (second half).
// One fused pass replaces the reference's FindCorrespondences (CameraTrackingUtils.cu:131-185),
// CalculateJacAndResKernel (Solver.cu:40-54) and the cublasSgemv / cublasSsyrk that reduce the
// 6 x N Jacobian to J^T r and J^T J (Solver.cpp:81-90): the Jacobian row of a pixel lives in
// registers and goes straight into the 27 running sums, so the 7.4 MB Jacobian matrix, the two
// correspondence maps and the residual map are never written (they are only on request, for the
// drop-in computeCorrespondences).  Oracle: oracle/vh_icp_oracle.c.
// Part of libvoxelhash_hip.so (gfx950); included by vh_kernels.hip after vh_device.h.
#pragma once

namespace vh {

constexpr int kIcpTerms = 29;        // 21 (upper triangle of J^T J) + 6 (J^T r) + sum d + count
constexpr int kIcpStride = 32;       // floats per partial record
constexpr int kIcpAbsDistance = 1;   // VH_ICP_ABS_DISTANCE
constexpr int kIcpNeedTarget = 2;    // VH_ICP_NEED_TARGET

struct IcpParams {
    float delta[12];     // rows 0..2 of the 4x4 that maps input points into the target's camera frame
    float K[9];          // row-major intrinsics (SetCameraIntrinsic, CameraTrackingUtils.cu:218-222)
    float distThres;
    int32_t width, height, flags;
};

// double -> int as cvt.rzi.s32.f64 (the reference's make_int2(double, double), :129): truncate,
// saturate, NaN -> 0; v_cvt_i32_f64 has the same contract.
__device__ __forceinline__ int d2i_rz(double x)
{
    int r;
    asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

// FindCorrespondences for one pixel.  Returns true when a correspondence is kept; t / n are the
// target point and normal, d the signed point-to-plane distance.
__device__ __forceinline__ bool icp_correspondence(const IcpParams &ip, const float4 *__restrict__ input,
                                                   const float4 *__restrict__ target,
                                                   const float4 *__restrict__ normals, int idx, float4 &t, float4 &n,
                                                   float &d)
{
    const float4 p = input[idx];
    if (p.z == 0.0f) return false;                                             // :148
    float q[3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
        q[r] = ip.delta[4 * r + 0] * p.x + ip.delta[4 * r + 1] * p.y + ip.delta[4 * r + 2] * p.z + ip.delta[4 * r + 3] * 1.0f;
    const float sx = ip.K[0] * q[0] + ip.K[1] * q[1] + ip.K[2] * q[2];       // cam2screenPos, :124-129
    const float sy = ip.K[3] * q[0] + ip.K[4] * q[1] + ip.K[5] * q[2];
    const float sz = ip.K[6] * q[0] + ip.K[7] * q[1] + ip.K[8] * q[2];
    const int u = d2i_rz((double)(sx / sz) + 0.5);
    const int v = d2i_rz((double)(sy / sz) + 0.5);
    if (!(u > 0 && v > 0 && u < ip.width && v < ip.height)) return false;     // :157 (strict > 0)
    const size_t ti = (size_t)v * ip.width + u;
    t = target[ti];
    n = normals[ti];
    if ((ip.flags & kIcpNeedTarget) && (t.z == 0.0f || (n.x == 0.0f && n.y == 0.0f && n.z == 0.0f))) return false;
    const float dx = q[0] - t.x, dy = q[1] - t.y, dz = q[2] - t.z;
    d = dx * n.x + dy * n.y + dz * n.z;                                        // :168-169
    return (ip.flags & kIcpAbsDistance) ? (__builtin_fabsf(d) < ip.distThres) : (d < ip.distThres);   // :170
}

// One pixel per lane; the 29 terms are reduced across the wave with shuffles, across the four
// waves through LDS, and each workgroup stores one partial record.  The order of the additions
// is fixed by the launch geometry, so the sums are reproducible run to run.
// corres / corresNormals / residuals: nullptr, or the maps computeCorrespondences fills.
__global__ __launch_bounds__(256) void icp_accumulate_kernel(const IcpParams ip, const float4 *__restrict__ input,
                                                             const float4 *__restrict__ target,
                                                             const float4 *__restrict__ normals,
                                                             float *__restrict__ partials,
                                                             float4 *__restrict__ corres,
                                                             float4 *__restrict__ corresNormals,
                                                             float *__restrict__ residuals)
{
    __shared__ float sm[4][kIcpStride];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    float acc[kIcpTerms];
#pragma unroll
    for (int k = 0; k < kIcpTerms; ++k) acc[k] = 0.0f;
    if (idx < ip.width * ip.height) {
        float4 t = make_float4(0.0f, 0.0f, 0.0f, 0.0f), n = t;
        float d = 0.0f;
        const bool kept = icp_correspondence(ip, input, target, normals, idx, t, n, d);
        if (kept) {
            // CalculateJacobians, Solver.cu:27-35: J = [n, target x n]
            const float J[6] = {n.x, n.y, n.z, t.y * n.z - t.z * n.y, t.z * n.x - t.x * n.z, t.x * n.y - t.y * n.x};
            int k = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int b = a; b < 6; ++b) acc[k++] = J[a] * J[b];
#pragma unroll
            for (int a = 0; a < 6; ++a) acc[21 + a] = J[a] * d;
            acc[27] = d;
            acc[28] = 1.0f;
        }
        if (corres) {                 // the reference clears the maps first (:198-200), then writes the kept ones
            const float4 zero = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            corres[idx] = kept ? t : zero;
            corresNormals[idx] = kept ? n : zero;
            residuals[idx] = kept ? d : 0.0f;
        }
    }
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < kIcpTerms; ++k) {
        float v = acc[k];
#pragma unroll
        for (int s = 1; s < kWave; s <<= 1) v += __shfl_xor(v, s);
        if (lane == 0) sm[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < kIcpStride)
        partials[(size_t)blockIdx.x * kIcpStride + threadIdx.x] =
            (threadIdx.x < kIcpTerms) ? ((sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x])) : 0.0f;
}

// One workgroup adds the partial records: lane (part, k) sums every 8th record, then the 8 parts.
__global__ __launch_bounds__(256) void icp_finalize_kernel(const float *__restrict__ partials, int32_t numBlocks,
                                                           float *__restrict__ out)
{
    __shared__ float sm[8][kIcpStride];
    const int k = threadIdx.x & (kIcpStride - 1), part = threadIdx.x >> 5;
    float s = 0.0f;
    for (int b = part; b < numBlocks; b += 8) s += partials[(size_t)b * kIcpStride + k];
    sm[part][k] = s;
    __syncthreads();
    if (threadIdx.x < kIcpStride) {
        float v = 0.0f;
#pragma unroll
        for (int p = 0; p < 8; ++p) v += sm[p][threadIdx.x];
        out[threadIdx.x] = v;
    }
}

// float depth image in metres -> vertex + normal maps: preProcess (CameraTrackingUtils.cu:50-113)
// without the /5000 of the uint16 path; turns a raycast depth image into an ICP target.
__device__ __forceinline__ float3 vertex_from_metres(const float *__restrict__ depth, const Mat3 &kinv, int W, int x,
                                                     int y)
{
    const float d = depth[(size_t)y * W + x];
    const float fx = (float)x, fy = (float)y;
    const float px = kinv.m[0] * fx + kinv.m[1] * fy + kinv.m[2] * 1.0f;
    const float py = kinv.m[3] * fx + kinv.m[4] * fy + kinv.m[5] * 1.0f;
    const float pz = kinv.m[6] * fx + kinv.m[7] * fy + kinv.m[8] * 1.0f;
    return make_float3(px * d, py * d, pz * d);
}

__global__ __launch_bounds__(256) void depth_to_maps_kernel(const float *__restrict__ depth, const Mat3 kinv, int W,
                                                            int H, float4 *__restrict__ positions,
                                                            float4 *__restrict__ normals)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= W * H) return;
    const int y = idx / W, x = idx - y * W;
    const float3 cc = vertex_from_metres(depth, kinv, W, x, y);
    positions[idx] = make_float4(cc.x, cc.y, cc.z, 1.0f);
    float4 n = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (x > 0 && x < W - 1 && y > 0 && y < H - 1) {
        const float3 pc = vertex_from_metres(depth, kinv, W, x, y + 1);
        const float3 cp = vertex_from_metres(depth, kinv, W, x + 1, y);
        const float3 mc = vertex_from_metres(depth, kinv, W, x, y - 1);
        const float3 cm = vertex_from_metres(depth, kinv, W, x - 1, y);
        if (cc.x != 0.0f && pc.x != 0.0f && cp.x != 0.0f && mc.x != 0.0f && cm.x != 0.0f) {
            const float ax = pc.x - mc.x, ay = pc.y - mc.y, az = pc.z - mc.z;
            const float bx = cp.x - cm.x, by = cp.y - cm.y, bz = cp.z - cm.z;
            const float nx = ay * bz - az * by, ny = az * bx - ax * bz, nz = ax * by - ay * bx;
            const float l = __builtin_sqrtf(nx * nx + ny * ny + nz * nz);
            if (l > 0.0f) n = make_float4(nx / l, ny / l, nz / l, 0.0f);
        }
    }
    normals[idx] = n;
}

}  // namespace vh
