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
constexpr int kIcpUnroll = 4;        // pixels a lane has in flight
constexpr int kIcpThreads = 256;     // lanes per workgroup (1024-lane workgroups measured slower: 20 vs 17 us per round)
constexpr int kIcpAbsDistance = 1;   // VH_ICP_ABS_DISTANCE
constexpr int kIcpNeedTarget = 2;    // VH_ICP_NEED_TARGET

struct IcpParams {
    float delta[12];     // rows 0..2 of the 4x4 that maps input points into the target's camera frame
    float K[9];          // row-major intrinsics (SetCameraIntrinsic, CameraTrackingUtils.cu:218-222)
    float distThres;
    int32_t width, height, flags;
};

// Device-resident state of a whole Align (vh_icp_align): the rounds chain on the stream without
// returning to the host; the solve of round i runs in the last workgroup of that round's launch.
struct IcpState {
    double T[16];        // the running estimate as a matrix, T = exp(estimate) (row-major)
    float delta[16];     // T in fp32: what the next round's pairing uses
    float sums[kIcpStride];   // the 29 sums of the last executed round
    int32_t rounds;      // rounds executed (systems built)
    int32_t done;        // 1: stop (summed residual exactly 0, CameraTracking.cpp:52, or singular system)
    int32_t singular;
    int32_t ticket;      // workgroups of the running round that have stored their partial record
};

// ---- SE3 (SE3.cpp:4-22): twist = (v, w), M = [[0,-w2,w1,v0],[w2,0,-w0,v1],[-w1,w0,0,v2],0]; the
// reference evaluates M.exp() / T.log() with Eigen's generic matrix functions, these are the closed
// forms of the same maps.  Shared by the host entry points and the device-side solve. ----
// Every loop below has a constant trip count and is unrolled, so that on the device the small
// matrices live in registers (left as loops they are indexed dynamically and go to scratch memory:
// the single-lane solve then cost 40 us per round instead of a few).
#define VH_UNROLL _Pragma("unroll")
__host__ __device__ inline void skew_terms(const double w[3], double Kx[9], double K2[9])
{
    Kx[0] = 0; Kx[1] = -w[2]; Kx[2] = w[1];
    Kx[3] = w[2]; Kx[4] = 0; Kx[5] = -w[0];
    Kx[6] = -w[1]; Kx[7] = w[0]; Kx[8] = 0;
    VH_UNROLL
    for (int i = 0; i < 3; ++i)
        VH_UNROLL
        for (int j = 0; j < 3; ++j)
            K2[3 * i + j] = Kx[3 * i] * Kx[j] + Kx[3 * i + 1] * Kx[3 + j] + Kx[3 * i + 2] * Kx[6 + j];
}

__host__ __device__ inline void se3_exp_d(const double twist[6], double T[16])
{
    const double *v = twist, *w = twist + 3;
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], th = sqrt(th2);
    double A, B, Cc;                       // sin(t)/t, (1-cos t)/t^2, (t-sin t)/t^3
    if (th < 1e-5) { A = 1.0 - th2 / 6.0; B = 0.5 - th2 / 24.0; Cc = 1.0 / 6.0 - th2 / 120.0; }
    else { A = sin(th) / th; B = (1.0 - cos(th)) / th2; Cc = (th - sin(th)) / (th2 * th); }
    double Kx[9], K2[9];
    skew_terms(w, Kx, K2);
    VH_UNROLL
    for (int i = 0; i < 16; ++i) T[i] = 0.0;
    VH_UNROLL
    for (int i = 0; i < 3; ++i) {
        double t = 0.0;
        VH_UNROLL
        for (int j = 0; j < 3; ++j) {
            const double I = (i == j) ? 1.0 : 0.0;
            T[4 * i + j] = I + A * Kx[3 * i + j] + B * K2[3 * i + j];
            t += (I + B * Kx[3 * i + j] + Cc * K2[3 * i + j]) * v[j];
        }
        T[4 * i + 3] = t;
    }
    T[15] = 1.0;
}

__host__ __device__ inline void se3_log_d(const double T[16], double twist[6])
{
    double c = 0.5 * (T[0] + T[5] + T[10] - 1.0);
    c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
    const double th = acos(c), th2 = th * th;
    const double r[3] = {T[9] - T[6], T[2] - T[8], T[4] - T[1]};            // (R - R^T) vee
    const double f = (th < 1e-5) ? 0.5 + th2 / 12.0 : th / (2.0 * sin(th));
    const double w[3] = {f * r[0], f * r[1], f * r[2]};
    const double D = (th < 1e-5) ? 1.0 / 12.0 + th2 / 720.0 : (1.0 - th * sin(th) / (2.0 * (1.0 - cos(th)))) / th2;
    double Kx[9], K2[9];
    skew_terms(w, Kx, K2);
    VH_UNROLL
    for (int i = 0; i < 3; ++i) {
        double s = 0.0;
        VH_UNROLL
        for (int j = 0; j < 3; ++j) s += (((i == j) ? 1.0 : 0.0) - 0.5 * Kx[3 * i + j] + D * K2[3 * i + j]) * T[4 * j + 3];
        twist[i] = s;
    }
    twist[3] = w[0]; twist[4] = w[1]; twist[5] = w[2];
}

// update = -(JTJ^-1 JTr) by Cholesky (Solver.cpp:104-105); false when JTJ is not positive definite.
__host__ __device__ inline bool icp_update_d(const double JTJ[36], const double JTr[6], double x[6])
{
    double L[36], y[6];
    VH_UNROLL
    for (int i = 0; i < 36; ++i) L[i] = 0.0;
    VH_UNROLL
    for (int i = 0; i < 6; ++i)
        VH_UNROLL
        for (int j = 0; j <= i; ++j) {
            double s = JTJ[6 * i + j];
            VH_UNROLL
            for (int k = 0; k < j; ++k) s -= L[6 * i + k] * L[6 * j + k];
            if (i == j) {
                if (!(s > 0.0)) return false;
                L[6 * i + i] = sqrt(s);
            } else {
                L[6 * i + j] = s / L[6 * j + j];
            }
        }
    VH_UNROLL
    for (int i = 0; i < 6; ++i) {
        double s = -JTr[i];
        VH_UNROLL
        for (int k = 0; k < i; ++k) s -= L[6 * i + k] * y[k];
        y[i] = s / L[6 * i + i];
    }
    VH_UNROLL
    for (int i = 5; i >= 0; --i) {
        double s = y[i];
        VH_UNROLL
        for (int k = i + 1; k < 6; ++k) s -= L[6 * k + i] * x[k];
        x[i] = s / L[6 * i + i];
    }
    return true;
}

// estimate = log(exp(update) exp(estimate))  (Solver.cpp:106); false (estimate untouched) when the
// system is singular.
__host__ __device__ inline bool icp_solve_d(const double JTJ[36], const double JTr[6], double estimate[6])
{
    double x[6], A[16], B[16], M[16];
    if (!icp_update_d(JTJ, JTr, x)) return false;
    se3_exp_d(x, A);
    se3_exp_d(estimate, B);
    VH_UNROLL
    for (int i = 0; i < 4; ++i)
        VH_UNROLL
        for (int j = 0; j < 4; ++j) {
            double s = 0.0;
            VH_UNROLL
            for (int k = 0; k < 4; ++k) s += A[4 * i + k] * B[4 * k + j];
            M[4 * i + j] = s;
        }
    se3_log_d(M, estimate);
    return true;
}

// The same step on the matrix itself, T <- exp(update) T: what exp(log(exp(update) exp(estimate)))
// evaluates to, without the logarithm and the second exponential (the device-side Align keeps T and
// takes the logarithm never; the single-lane solve is on the critical path of every round).
__host__ __device__ inline bool icp_step_matrix_d(const double JTJ[36], const double JTr[6], double T[16])
{
    double x[6], A[16], M[16];
    if (!icp_update_d(JTJ, JTr, x)) return false;
    se3_exp_d(x, A);
    VH_UNROLL
    for (int i = 0; i < 4; ++i)
        VH_UNROLL
        for (int j = 0; j < 4; ++j) {
            double s = 0.0;
            VH_UNROLL
            for (int k = 0; k < 4; ++k) s += A[4 * i + k] * T[4 * k + j];
            M[4 * i + j] = s;
        }
    VH_UNROLL
    for (int i = 0; i < 16; ++i) T[i] = M[i];
    return true;
}
#undef VH_UNROLL

// double -> int as cvt.rzi.s32.f64 (the reference's make_int2(double, double), :129): truncate,
// saturate, NaN -> 0; v_cvt_i32_f64 has the same contract.
__device__ __forceinline__ int d2i_rz(double x)
{
    int r;
    asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

// FindCorrespondences for one pixel, in two halves so that a lane can have the gathers of several
// pixels in flight:
//   icp_project: moved point q and the target pixel it lands on (false: no pairing possible)
//   icp_pair:    signed point-to-plane distance d against the gathered target point t and normal n,
//                and whether the pair is kept
__device__ __forceinline__ bool icp_project(const IcpParams &ip, const float4 p, float q[3], int &ti)
{
    if (p.z == 0.0f) return false;                                             // :148
#pragma unroll
    for (int r = 0; r < 3; ++r)
        q[r] = ip.delta[4 * r + 0] * p.x + ip.delta[4 * r + 1] * p.y + ip.delta[4 * r + 2] * p.z + ip.delta[4 * r + 3] * 1.0f;
    const float sx = ip.K[0] * q[0] + ip.K[1] * q[1] + ip.K[2] * q[2];       // cam2screenPos, :124-129
    const float sy = ip.K[3] * q[0] + ip.K[4] * q[1] + ip.K[5] * q[2];
    const float sz = ip.K[6] * q[0] + ip.K[7] * q[1] + ip.K[8] * q[2];
    const int u = d2i_rz((double)(sx / sz) + 0.5);
    const int v = d2i_rz((double)(sy / sz) + 0.5);
    if (!(u > 0 && v > 0 && u < ip.width && v < ip.height)) return false;     // :157 (strict > 0)
    ti = v * ip.width + u;
    return true;
}

__device__ __forceinline__ bool icp_pair(const IcpParams &ip, const float q[3], const float4 t, const float4 n, float &d)
{
    if ((ip.flags & kIcpNeedTarget) && (t.z == 0.0f || (n.x == 0.0f && n.y == 0.0f && n.z == 0.0f))) return false;
    const float dx = q[0] - t.x, dy = q[1] - t.y, dz = q[2] - t.z;
    d = dx * n.x + dy * n.y + dz * n.z;                                        // :168-169
    return (ip.flags & kIcpAbsDistance) ? (__builtin_fabsf(d) < ip.distThres) : (d < ip.distThres);   // :170
}

// Sum over the 64 lanes of a wave, result in lane 63: an inclusive scan in DPP steps (row_shr 1, 2,
// 4, 8 inside each row of 16 lanes, then row_bcast:15 and row_bcast:31 carry the row totals on), six
// VALU instructions and no LDS traffic per term; __shfl_xor goes through ds_bpermute and made the 29
// reductions a visible part of the round.
template <int kCtrl, int kRowMask>
__device__ __forceinline__ float dpp_take(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), kCtrl, kRowMask, 0xf, false));
}

__device__ __forceinline__ float wave_sum_lane63(float v)
{
    v += dpp_take<0x111, 0xf>(v);
    v += dpp_take<0x112, 0xf>(v);
    v += dpp_take<0x114, 0xf>(v);
    v += dpp_take<0x118, 0xf>(v);
    v += dpp_take<0x142, 0xa>(v);
    v += dpp_take<0x143, 0xc>(v);
    return v;
}

// One launch per round.  256-lane workgroups (one per compute unit) stride over the pixels and keep the 29 terms in
// registers; each wave reduces them with DPP adds, the four waves combine through LDS and the
// workgroup stores one partial record.  The last workgroup to finish (ticket) adds the records in a
// fixed order -- so the sums are reproducible run to run -- and, when `solve` is set, lane 0 solves
// the 6x6 system in double and advances the estimate in the device-resident state, so the next
// round starts without the host (the reference returns to the host three times per round,
// Solver.cpp:83,89 and CameraTrackingUtils.cu:212).
//   useState: take the estimate from state->delta (Align) instead of ip.delta (step API)
//   corres / corresNormals / residuals: nullptr, or the maps computeCorrespondences fills
template <bool kWriteMaps>
__global__ __launch_bounds__(kIcpThreads) void icp_round_kernel(IcpParams ip, const float4 *__restrict__ input,
                                                        const float4 *__restrict__ target,
                                                        const float4 *__restrict__ normals,
                                                        float *__restrict__ partials, float4 *__restrict__ corres,
                                                        float4 *__restrict__ corresNormals,
                                                        float *__restrict__ residuals, IcpState *__restrict__ state,
                                                        int useState, int solve)
{
    __shared__ float sm[kIcpThreads / kWave][kIcpStride];
    __shared__ float total[kIcpStride];
    __shared__ int isLast;
    if (useState) {
        if (state->done) return;
#pragma unroll
        for (int i = 0; i < 12; ++i) ip.delta[i] = state->delta[i];
    }
    float acc[kIcpTerms];
#pragma unroll
    for (int k = 0; k < kIcpTerms; ++k) acc[k] = 0.0f;
    const int npix = ip.width * ip.height;
    // kIcpUnroll pixels per pass: their input points are loaded together, then their target points and
    // normals are gathered together, then the pairs are accumulated (one memory latency per stage for
    // the group instead of one per pixel)
    const int stride = gridDim.x * kIcpThreads;
    for (int base = blockIdx.x * kIcpThreads + threadIdx.x; base < npix; base += stride * kIcpUnroll) {
        float4 p[kIcpUnroll], t[kIcpUnroll], n[kIcpUnroll];
        float q[kIcpUnroll][3];
        int ti[kIcpUnroll];
        bool ok[kIcpUnroll];
#pragma unroll
        for (int u = 0; u < kIcpUnroll; ++u) {
            const int idx = base + u * stride;
            p[u] = idx < npix ? input[idx] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
#pragma unroll
        for (int u = 0; u < kIcpUnroll; ++u) {
            ti[u] = 0;
            ok[u] = (base + u * stride < npix) && icp_project(ip, p[u], q[u], ti[u]);
        }
#pragma unroll
        for (int u = 0; u < kIcpUnroll; ++u) {
            t[u] = target[ti[u]];                  // pixel 0 when there is no pairing: harmless, unused
            n[u] = normals[ti[u]];
        }
#pragma unroll
        for (int u = 0; u < kIcpUnroll; ++u) {
            const int idx = base + u * stride;
            float d = 0.0f;
            const bool kept = ok[u] && icp_pair(ip, q[u], t[u], n[u], d);
            if (kept) {
                // CalculateJacobians, Solver.cu:27-35: J = [n, target x n]
                const float4 tt = t[u], nn = n[u];
                const float J[6] = {nn.x, nn.y, nn.z, tt.y * nn.z - tt.z * nn.y, tt.z * nn.x - tt.x * nn.z,
                                    tt.x * nn.y - tt.y * nn.x};
                int k = 0;
#pragma unroll
                for (int a = 0; a < 6; ++a)
#pragma unroll
                    for (int b = a; b < 6; ++b) acc[k++] += J[a] * J[b];
#pragma unroll
                for (int a = 0; a < 6; ++a) acc[21 + a] += J[a] * d;
                acc[27] += d;
                acc[28] += 1.0f;
            }
            if constexpr (kWriteMaps) {   // the reference clears the maps first (:198-200), then writes the kept ones
                if (idx < npix) {
                    const float4 zero = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    corres[idx] = kept ? t[u] : zero;
                    corresNormals[idx] = kept ? n[u] : zero;
                    residuals[idx] = kept ? d : 0.0f;
                }
            }
        }
    }
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < kIcpTerms; ++k) {
        const float v = wave_sum_lane63(acc[k]);
        if (lane == kWave - 1) sm[wave][k] = v;
    }
    __syncthreads();
    // the record is stored, released and ticketed by wave 0 alone: an agent-scope release writes the
    // L2 of this XCD back, and executed by all 256 lanes of 512 workgroups it cost 40 us per round
    if (wave == 0) {
        if (threadIdx.x < kIcpStride) {
            float v = 0.0f;
#pragma unroll
            for (int w = 0; w < kIcpThreads / kWave; ++w) v += sm[w][threadIdx.x];
            // (a device-scope store: coherent across the XCDs by itself)
            __hip_atomic_store(&partials[(size_t)blockIdx.x * kIcpStride + threadIdx.x], (threadIdx.x < kIcpTerms) ? v : 0.0f,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // the record is complete before the ticket is drawn: the wave's stores have been acknowledged (s_waitcnt), no cache
        // write-back -- round 4: an agent-scope release here is an L2 write-back per workgroup, which the chip serves one at
        // a time (vh_icp_align of 20 rounds, same box: 350 us with __threadfence() here, 312 us with this)
        // This hand-off is NOT a release / acquire pair of the HIP memory model: it rests on gfx942 / gfx950 behaviour -- an
        // agent-scope relaxed atomic store is a write-through (sc1) store, complete for every XCD once s_waitcnt has seen it
        // acknowledged, and the reader's agent-scope atomic loads (sc1) never hit a stale line of their own L2
        // (MI355X_MICROARCH.md, valid forms: "every store of the handed-off bytes sc1 and drained before the counter, every
        // load of them an sc1 load").  RULE: every store a workgroup makes before its ticket must be such an atomic store; a
        // plain store added here, or another architecture (a separate store counter), needs __threadfence() back.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "vh_icp.hip: the ticket hand-off without a cache write-back is only valid on gfx942 / gfx950 (see the comment above)"
#endif
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        if (threadIdx.x == 0) isLast = atomicAdd(&state->ticket, 1) == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (!isLast) return;

    // last workgroup: lane (part, k) of the first 256 adds every 8th record, then the 8 parts.  The
    // records are read 16 at a time into registers so that the loads are in flight together (one
    // dependent load after the other cost 40 us here); they were written by other compute units
    // before their ticket, and this workgroup's L1 has never held them.
    const int k = threadIdx.x & (kIcpStride - 1), part = threadIdx.x >> 5;
    const int numRecords = (int)gridDim.x;
    if (threadIdx.x < 256) {
        float s = 0.0f;
        for (int b0 = part; b0 < numRecords; b0 += 8 * 16) {
            float r[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int b = b0 + 8 * u;
                r[u] = (b < numRecords) ? __hip_atomic_load(&partials[(size_t)b * kIcpStride + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) s += r[u];
        }
        sm[part][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < kIcpStride) {
        float v = 0.0f;
#pragma unroll
        for (int p = 0; p < 8; ++p) v += sm[p][threadIdx.x];
        total[threadIdx.x] = v;
        state->sums[threadIdx.x] = v;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    state->ticket = 0;
    state->rounds += 1;
    if (!solve) return;
    if (total[27] == 0.0f) { state->done = 1; return; }                 // CameraTracking.cpp:52
    double JTJ[36], JTr[6], T[16];
    int t = 0;
    for (int a = 0; a < 6; ++a)
        for (int b = a; b < 6; ++b) {
            JTJ[6 * a + b] = JTJ[6 * b + a] = (double)total[t];
            ++t;
        }
    for (int a = 0; a < 6; ++a) JTr[a] = (double)total[21 + a];
    for (int i = 0; i < 16; ++i) T[i] = state->T[i];
    if (!icp_step_matrix_d(JTJ, JTr, T)) { state->done = 1; state->singular = 1; return; }
    for (int i = 0; i < 16; ++i) { state->T[i] = T[i]; state->delta[i] = (float)T[i]; }
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
