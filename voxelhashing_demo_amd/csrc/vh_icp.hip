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
    int32_t ticket;      // workgroups of the running round that have stored their partial record (icp_round_kernel)
    int32_t timeout;     // a workgroup of the one-launch Align gave up waiting (vh_icp_align reports VH_ERR_TIMEOUT)
    int32_t pad[3];
};

// What the one-launch Align starts from, by value in the kernel arguments (no copy in front of the launch).
struct IcpStart {
    double T[16];
};

// ---- SE3 (SE3.cpp:4-22): twist = (v, w), M = [[0,-w2,w1,v0],[w2,0,-w0,v1],[-w1,w0,0,v2],0]; the
// reference evaluates M.exp() / T.log() with Eigen's generic matrix functions, these are the closed
// forms of the same maps.  Shared by the host entry points and the device-side solve. ----
// Every loop below has a constant trip count and is unrolled, so that on the device the small
// matrices live in registers (left as loops they are indexed dynamically and go to scratch memory:
// the single-lane solve then cost 40 us per round instead of a few).
#define VH_UNROLL _Pragma("unroll")
// The solve's multiply-adds are fused on purpose (the library is built -ffp-contract=off for the model's arithmetic): on the
// device they run on ONE lane between two rounds of Align at half-rate double issue, and a fused multiply-add is one
// instruction where a product and a sum are two; host and device share these functions, so they agree bit for bit.
#define VH_FMA(a, b, c) __builtin_fma((a), (b), (c))
__host__ __device__ inline void skew_terms(const double w[3], double Kx[9], double K2[9])
{
    Kx[0] = 0; Kx[1] = -w[2]; Kx[2] = w[1];
    Kx[3] = w[2]; Kx[4] = 0; Kx[5] = -w[0];
    Kx[6] = -w[1]; Kx[7] = w[0]; Kx[8] = 0;
    VH_UNROLL
    for (int i = 0; i < 3; ++i)
        VH_UNROLL
        for (int j = 0; j < 3; ++j)
            K2[3 * i + j] = VH_FMA(Kx[3 * i + 2], Kx[6 + j], VH_FMA(Kx[3 * i + 1], Kx[3 + j], Kx[3 * i] * Kx[j]));
}

__host__ __device__ inline void se3_exp_d(const double twist[6], double T[16])
{
    const double *v = twist, *w = twist + 3;
    const double th2 = VH_FMA(w[2], w[2], VH_FMA(w[1], w[1], w[0] * w[0]));
    double A, B, Cc;                       // sin(t)/t, (1-cos t)/t^2, (t-sin t)/t^3: even functions of t
    if (th2 < 0.25) {
        // their power series in t^2, nine terms (the tenth is below 2^-64 of the first for t < 0.5): an ICP update is a small
        // rotation, and on the device this runs on one lane between two rounds of Align -- no square root, no sin / cos
        // (some 400 double instructions) and none of the three divisions of the closed forms
        // 1/(2n+1)!, 1/(2n+2)!, 1/(2n+3)! for n = 0..8, alternating signs, Horner in t^2
        const double fA[9] = {1.0, 1.0 / 6, 1.0 / 120, 1.0 / 5040, 1.0 / 362880, 1.0 / 39916800, 1.0 / 6227020800.0,
                              1.0 / 1307674368000.0, 1.0 / 355687428096000.0};
        const double fB[9] = {1.0 / 2, 1.0 / 24, 1.0 / 720, 1.0 / 40320, 1.0 / 3628800, 1.0 / 479001600, 1.0 / 87178291200.0,
                              1.0 / 20922789888000.0, 1.0 / 6402373705728000.0};
        const double fC[9] = {1.0 / 6, 1.0 / 120, 1.0 / 5040, 1.0 / 362880, 1.0 / 39916800, 1.0 / 6227020800.0,
                              1.0 / 1307674368000.0, 1.0 / 355687428096000.0, 1.0 / 121645100408832000.0};
        A = fA[8]; B = fB[8]; Cc = fC[8];
        VH_UNROLL
        for (int n = 7; n >= 0; --n) {
            A = VH_FMA(-th2, A, fA[n]);
            B = VH_FMA(-th2, B, fB[n]);
            Cc = VH_FMA(-th2, Cc, fC[n]);
        }
    } else {
        const double th = sqrt(th2);
        A = sin(th) / th; B = (1.0 - cos(th)) / th2; Cc = (th - sin(th)) / (th2 * th);
    }
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
            T[4 * i + j] = VH_FMA(B, K2[3 * i + j], VH_FMA(A, Kx[3 * i + j], I));
            t = VH_FMA(VH_FMA(Cc, K2[3 * i + j], VH_FMA(B, Kx[3 * i + j], I)), v[j], t);
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

// update = -(JTJ^-1 JTr) (Solver.cpp:104-105) by an LDL^T factorisation; false when JTJ is not positive definite (a pivot
// <= 0: the same criterion as a Cholesky factorisation's, whose pivots are these).  Six divisions and no square root -- the
// solve runs on ONE lane between two rounds of Align, so every IEEE double division (~15 dependent instructions) and square
// root on its chain is paid by the whole chip (as Cholesky: 6 square roots, 27 divisions; a 20-round Align 326 -> 296 us).
__host__ __device__ inline bool icp_update_d(const double JTJ[36], const double JTr[6], double x[6])
{
    double L[36], Wd[36], inv[6], y[6];      // L unit lower triangular, Wd[i][k] = L[i][k] d[k], inv[k] = 1 / d[k]
    VH_UNROLL
    for (int i = 0; i < 36; ++i) { L[i] = 0.0; Wd[i] = 0.0; }
    VH_UNROLL
    for (int j = 0; j < 6; ++j) {
        double s = JTJ[6 * j + j];
        VH_UNROLL
        for (int k = 0; k < j; ++k) s = VH_FMA(-Wd[6 * j + k], L[6 * j + k], s);
        if (!(s > 0.0)) return false;
        inv[j] = 1.0 / s;
        VH_UNROLL
        for (int i = j + 1; i < 6; ++i) {
            double t = JTJ[6 * i + j];
            VH_UNROLL
            for (int k = 0; k < j; ++k) t = VH_FMA(-Wd[6 * i + k], L[6 * j + k], t);
            Wd[6 * i + j] = t;
            L[6 * i + j] = t * inv[j];
        }
    }
    VH_UNROLL
    for (int i = 0; i < 6; ++i) {              // L y = -JTr
        double s = -JTr[i];
        VH_UNROLL
        for (int k = 0; k < i; ++k) s = VH_FMA(-L[6 * i + k], y[k], s);
        y[i] = s;
    }
    VH_UNROLL
    for (int i = 5; i >= 0; --i) {             // L^T x = D^-1 y
        double s = y[i] * inv[i];
        VH_UNROLL
        for (int k = i + 1; k < 6; ++k) s = VH_FMA(-L[6 * k + i], x[k], s);
        x[i] = s;
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
            for (int k = 0; k < 4; ++k) s = VH_FMA(A[4 * i + k], T[4 * k + j], s);
            M[4 * i + j] = s;
        }
    VH_UNROLL
    for (int i = 0; i < 16; ++i) T[i] = M[i];
    return true;
}
#undef VH_FMA
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

// The workgroup's sum of each of the 29 terms.  Every lane leaves its terms in LDS ([term][lane], rows 260 floats apart: the
// eight terms a wave reads then start in different banks), eight lanes per term add 32 values each (float4 reads, lane `sub`
// of a term takes every 8th group of four) and meet in three DPP adds (xor 1, xor 2, mirror of 8) -- ~80 instructions per lane.
// (Rounds 2-5 reduced each term across the wave with six DPP adds and the four waves through LDS: 29 x 6 DPP steps with their
// wait states, 430 instructions as compiled, 1.5 us of a round.)  Returns to lanes 8t .. 8t + 7 the sum of term t (t < 29; 0
// beyond).  A fixed order of additions: the sums are reproducible run to run.
constexpr int kIcpRow = kIcpThreads + 4;
constexpr int kIcpSumFloats = kIcpTerms * kIcpRow;

template <int kCtrl>
__device__ __forceinline__ float dpp_take(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), kCtrl, 0xf, 0xf, false));
}

__device__ __forceinline__ float icp_workgroup_sums(const float acc[kIcpTerms], float *lds)
{
#pragma unroll
    for (int k = 0; k < kIcpTerms; ++k) lds[k * kIcpRow + threadIdx.x] = acc[k];
    __syncthreads();
    const int term = threadIdx.x >> 3, sub = threadIdx.x & 7;
    float v = 0.0f;
    if (term < kIcpTerms) {
        const float4 *row = reinterpret_cast<const float4 *>(lds + term * kIcpRow);
#pragma unroll
        for (int j = 0; j < kIcpThreads / 32; ++j) {
            const float4 x = row[8 * j + sub];
            v += x.x; v += x.y; v += x.z; v += x.w;
        }
    }
    v += dpp_take<0xB1>(v);        // quad_perm [1,0,3,2]
    v += dpp_take<0x4E>(v);        // quad_perm [2,3,0,1]
    v += dpp_take<0x141>(v);       // row_half_mirror
    return v;
}

// CalculateJacobians, Solver.cu:27-35: J = [n, target x n]; the row's 21 + 6 products, d and 1 into the running sums
// (measured: fused multiply-adds here -- 27 scalar v_fmac instead of the packed multiplies and adds the compiler forms -- made
// the round's pixel pass slower, 2.64 against 2.36 us)
__device__ __forceinline__ void icp_accumulate(float acc[kIcpTerms], const float4 tt, const float4 nn, const float d)
{
    const float J[6] = {nn.x, nn.y, nn.z, tt.y * nn.z - tt.z * nn.y, tt.z * nn.x - tt.x * nn.z, tt.x * nn.y - tt.y * nn.x};
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

// The workgroup's 29 sums -> one partial record, then a ticket from `ticket`; returns the ticket to every lane.
__device__ __forceinline__ int icp_store_record(const float acc[kIcpTerms], float *lds, int *drawn,
                                                float *__restrict__ partials, int32_t *ticket)
{
    const float v = icp_workgroup_sums(acc, lds);
    // (a device-scope store: coherent across the XCDs by itself)
    if ((threadIdx.x & 7) == 0)
        __hip_atomic_store(&partials[(size_t)blockIdx.x * kIcpStride + (threadIdx.x >> 3)], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // the record is complete before the ticket is drawn: every wave's stores have been acknowledged (s_waitcnt), no cache
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
    __syncthreads();
    if (threadIdx.x == 0) *drawn = atomicAdd(ticket, 1);
    __syncthreads();
    return *drawn;
}

// The workgroup that drew the last ticket adds the records in a fixed order -- so the sums are reproducible run to run:
// lane (part, k) of the 256 adds every 8th record, then the 8 parts; total[0..31] in LDS for every lane afterwards.  The
// records are read 16 at a time into registers so that the loads are in flight together (one dependent load after the
// other cost 40 us here); they were written by other compute units before their ticket, and this workgroup's L1 has never
// held them.
__device__ __forceinline__ void icp_sum_records(const float *__restrict__ partials, const int numRecords,
                                                float (*sm)[kIcpStride], float *total)
{
    const int k = threadIdx.x & (kIcpStride - 1), part = threadIdx.x >> 5;
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
    __syncthreads();
    if (threadIdx.x < kIcpStride) {
        float v = 0.0f;
#pragma unroll
        for (int p = 0; p < 8; ++p) v += sm[p][threadIdx.x];
        total[threadIdx.x] = v;
    }
    __syncthreads();
}

// The 6x6 system from the 29 sums and one step of the estimate, T <- exp(update) T; false: singular.
__device__ __forceinline__ bool icp_step_from_sums(const float *total, double T[16])
{
    double JTJ[36], JTr[6];
    int t = 0;
    for (int a = 0; a < 6; ++a)
        for (int b = a; b < 6; ++b) {
            JTJ[6 * a + b] = JTJ[6 * b + a] = (double)total[t];
            ++t;
        }
    for (int a = 0; a < 6; ++a) JTr[a] = (double)total[21 + a];
    return icp_step_matrix_d(JTJ, JTr, T);
}

// One launch per round.  256-lane workgroups (one per compute unit) stride over the pixels and keep the 29 terms in
// registers; every workgroup stores one partial record (icp_store_record); the last one to finish adds the records
// (icp_sum_records) and, when `solve` is set, its lane 0 solves the 6x6 system in double and advances the estimate in the
// device-resident state, so the next round starts without the host (the reference returns to the host three times per
// round, Solver.cpp:83,89 and CameraTrackingUtils.cu:212).  The step API (vh_icp_build_system, computeCorrespondences)
// and the Align of images too large for icp_align_kernel run on it.
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
    __shared__ float4 sums4[kIcpSumFloats / 4];
    __shared__ float sm[8][kIcpStride];
    __shared__ float total[kIcpStride];
    __shared__ int drawn;
    float *sums = reinterpret_cast<float *>(sums4);
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
            if (kept) icp_accumulate(acc, t[u], n[u], d);
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
    if (icp_store_record(acc, sums, &drawn, partials, &state->ticket) != (int)gridDim.x - 1) return;
    icp_sum_records(partials, (int)gridDim.x, sm, total);
    if (threadIdx.x < kIcpStride) state->sums[threadIdx.x] = total[threadIdx.x];
    if (threadIdx.x != 0) return;
    state->ticket = 0;
    state->rounds += 1;
    if (!solve) return;
    if (total[27] == 0.0f) { state->done = 1; return; }                 // CameraTracking.cpp:52
    double T[16];
    for (int i = 0; i < 16; ++i) T[i] = state->T[i];
    if (!icp_step_from_sums(total, T)) { state->done = 1; state->singular = 1; return; }
    for (int i = 0; i < 16; ++i) { state->T[i] = T[i]; state->delta[i] = (float)T[i]; }
}

// ---- All rounds of an Align in ONE launch (vh_icp_align).  The chain of one-launch rounds pays per round a launch gap and
// ramp, the input points again, 256 tickets drawn from one address and the solve between two launches; here the grid
// stays: a lane keeps its kSlots input points in registers for all rounds (the same pixels, the same order of additions and
// the same per-workgroup sums as icp_round_kernel with this grid, added in the same order: bit-identical to the chain,
// tests/test_gpu_icp.py), and nothing is counted -- every 8-byte word handed between workgroups carries its own sequence
// number {value : float, seq : int}, written and read as ONE 64-bit agent-scope access, so a reader that finds the
// round's number has that word after one round trip (no store-then-flag, no flag-then-load: each would be a dependent trip):
//   project + gather (the maps sit in the XCD's L2 from round 1 on) -> the workgroup's 29 sums as 29 such words ->
//   workgroup 0 polls all records (lane (part, k) term k of every 8th record: icp_sum_records' partition), adds, its lane 0
//   solves and publishes the next estimate as 12 such words (eight copies, a copy per blockIdx & 7, so that the polls of
//   256 workgroups do not meet on one memory channel) -> everyone polls its copy.
// Sequence numbers grow over the life of the vh_icp (seqBase), so nothing is ever reset.  A grid-wide wait: every workgroup
// must be able to be resident (vh_icp_create checks the grid against what the chip holds at this kernel's occupancy) and every poll
// loop is bounded -- workgroup 0 giving up publishes the stop itself, another one sets state->timeout and leaves; the call
// returns VH_ERR_TIMEOUT.
constexpr int kIcpPubCopies = 8, kIcpPubStride = 16;      // words per copy (12 used; 128 bytes apart)

__device__ __forceinline__ unsigned long long icp_word(const float v, const int seq)
{
    return (unsigned long long)__builtin_bit_cast(unsigned, v) | ((unsigned long long)(unsigned)seq << 32);
}
__device__ __forceinline__ unsigned long long icp_load_word(const unsigned long long *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void icp_store_word(unsigned long long *p, const unsigned long long w)
{
    __hip_atomic_store(p, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int kSlots>
__global__ __launch_bounds__(kIcpThreads) void icp_align_kernel(IcpParams ip, const float4 *__restrict__ input,
                                                                const float4 *__restrict__ target,
                                                                const float4 *__restrict__ normals,
                                                                unsigned long long *__restrict__ records,
                                                                unsigned long long *__restrict__ pub,
                                                                const IcpStart start, IcpState *__restrict__ state, const int maxIters,
                                                                const int seqBase, const uint32_t spinLimit,
                                                                unsigned long long *__restrict__ stamps)
{
    // diagnostics (VH_ICP_STAMPS=1): s_memrealtime (100 MHz) per round, workgroup 0: [0] round starts, [1] sums in registers,
    // [2] record stored, [3] all records seen, [4] added, [5] next estimate published; the last workgroup: [6] round starts, [7] record stored
#define VH_ICP_STAMP(i) do { if (stamps && blockIdx.x == 0 && threadIdx.x == 0) stamps[(size_t)round * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
    __shared__ float4 sums4[kIcpSumFloats / 4];
    __shared__ float sm[8][kIcpStride];
    __shared__ float total[kIcpStride];
    __shared__ double sT[16];            // workgroup 0: the estimate in double
    __shared__ float sDelta[12];         // rows 0..2 of the estimate the running round uses
    __shared__ int go, seenBy;
    float *sums = reinterpret_cast<float *>(sums4);
    const int npix = ip.width * ip.height, stride = gridDim.x * kIcpThreads, base = blockIdx.x * kIcpThreads + threadIdx.x;
    const int numBlocks = (int)gridDim.x, lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    // kSlots > 0: a lane's kSlots input points stay in registers for all rounds; kSlots == 0 (images too large for that): the
    // pixel pass is icp_round_kernel's strided loop, the points read again every round
    float4 p[kSlots > 0 ? kSlots : 1];
#pragma unroll
    for (int u = 0; u < kSlots; ++u) p[u] = (base + u * stride < npix) ? input[base + u * stride] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (threadIdx.x < 16) {
        const double t = start.T[threadIdx.x];
        sT[threadIdx.x] = t;
        if (threadIdx.x < 12) sDelta[threadIdx.x] = (float)t;
    }
    __syncthreads();
    for (int round = 0; round < maxIters; ++round) {
        const int want = seqBase + round + 1;
        const bool finalRound = round == maxIters - 1;
#pragma unroll
        for (int i = 0; i < 12; ++i) ip.delta[i] = sDelta[i];
        VH_ICP_STAMP(0);
        if (stamps && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) stamps[(size_t)round * 8 + 6] = __builtin_amdgcn_s_memrealtime();
        float acc[kIcpTerms];
#pragma unroll
        for (int k = 0; k < kIcpTerms; ++k) acc[k] = 0.0f;
        if constexpr (kSlots > 0) {
            float4 t[kSlots], n[kSlots];
            float q[kSlots][3];
            int ti[kSlots];
            bool ok[kSlots];
#pragma unroll
            for (int u = 0; u < kSlots; ++u) {
                ti[u] = 0;
                ok[u] = (base + u * stride < npix) && icp_project(ip, p[u], q[u], ti[u]);
            }
#pragma unroll
            for (int u = 0; u < kSlots; ++u) {
                t[u] = target[ti[u]];
                n[u] = normals[ti[u]];
            }
#pragma unroll
            for (int u = 0; u < kSlots; ++u) {
                float d = 0.0f;
                if (ok[u] && icp_pair(ip, q[u], t[u], n[u], d)) icp_accumulate(acc, t[u], n[u], d);
            }
        } else {
            for (int b = base; b < npix; b += stride * kIcpUnroll) {
                float4 pp[kIcpUnroll], t[kIcpUnroll], n[kIcpUnroll];
                float q[kIcpUnroll][3];
                int ti[kIcpUnroll];
                bool ok[kIcpUnroll];
#pragma unroll
                for (int u = 0; u < kIcpUnroll; ++u) pp[u] = b + u * stride < npix ? input[b + u * stride] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
                for (int u = 0; u < kIcpUnroll; ++u) {
                    ti[u] = 0;
                    ok[u] = (b + u * stride < npix) && icp_project(ip, pp[u], q[u], ti[u]);
                }
#pragma unroll
                for (int u = 0; u < kIcpUnroll; ++u) {
                    t[u] = target[ti[u]];
                    n[u] = normals[ti[u]];
                }
#pragma unroll
                for (int u = 0; u < kIcpUnroll; ++u) {
                    float d = 0.0f;
                    if (ok[u] && icp_pair(ip, q[u], t[u], n[u], d)) icp_accumulate(acc, t[u], n[u], d);
                }
            }
        }
        VH_ICP_STAMP(1);
        {
            const float v = icp_workgroup_sums(acc, sums);
            if ((threadIdx.x & 7) == 0 && threadIdx.x < 8 * kIcpTerms)
                icp_store_word(&records[(size_t)blockIdx.x * kIcpStride + (threadIdx.x >> 3)], icp_word(v, want));
        }
        VH_ICP_STAMP(2);
        if (stamps && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) stamps[(size_t)round * 8 + 7] = __builtin_amdgcn_s_memrealtime();
        if (stamps && round == 10 && threadIdx.x == 0) stamps[512 + blockIdx.x] = __builtin_amdgcn_s_memrealtime();
        if (blockIdx.x != 0) {
            if (finalRound) return;
            // wait for the estimate of the next round (or the stop): the copy of this workgroup's octet
            // all four waves poll, a quarter of a round trip apart (the first to see it tells the others through LDS): the
            // estimate is noticed a quarter of a poll after it lands, not half of one
            {
                const unsigned long long *word = pub + (size_t)(blockIdx.x & (kIcpPubCopies - 1)) * kIcpPubStride + (lane < 12 ? lane : 0);
                if (threadIdx.x == 0) seenBy = 0;
                __builtin_amdgcn_s_sleep(64);         // (nothing can be there before the slowest record, the sum and the solve)
                __syncthreads();
                for (int i = 0; i < wave; ++i) __builtin_amdgcn_s_sleep(8);
                int verdict = 0;                      // 0 timeout, 1 go, -1 stop
                for (uint32_t it = 0; it < spinLimit; ++it) {
                    const unsigned long long w = icp_load_word(word);
                    const int seq = (int)(w >> 32);
                    if (__builtin_amdgcn_ballot_w64(seq != want && seq != -want) == 0ull) {
                        verdict = seq > 0 ? 1 : -1;
                        if (lane < 12) sDelta[lane] = __builtin_bit_cast(float, (unsigned)w);      // (every wave that sees it writes the same words)
                        if (lane == 0) { go = verdict == 1; __hip_atomic_store(&seenBy, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
                        break;
                    }
                    if (__hip_atomic_load(&seenBy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) { verdict = 2; break; }
                }
                if (verdict == 0 && lane == 0) {
                    __hip_atomic_store(&state->timeout, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    go = 0;
                }
            }
            __syncthreads();
            if (!go) return;
            continue;
        }
        // ---- workgroup 0: all records of this round, added in icp_sum_records' order
        const int k = threadIdx.x & (kIcpStride - 1), part = threadIdx.x >> 5;
        const int kk = k < kIcpTerms ? k : 0;               // (lanes 29..31 of a part ride along on term 0)
        float s = 0.0f;
        bool timedOut = false;
        // The workgroups finish a round within ~0.5 us of each other and a stored word takes ~0.7 us to be visible: a first pass
        // started at once finds most words missing and only delays the pass that counts -- a short nap first (s_sleep 16, ~0.4 us:
        // period of a round 8.3 us; 8: 8.9, 12: 8.3 - 8.8, 20: 8.45, 24: 8.55, none: 8.9; tools/r06_icp_ab.sh)
#ifndef VH_ICP_NAP
#define VH_ICP_NAP 16
#endif
        __builtin_amdgcn_s_sleep(VH_ICP_NAP);
        for (int b0 = part; b0 < numBlocks; b0 += 8 * 32) {
            // a word that has been seen is kept (in registers: parked in LDS the round took 9.0 instead of 8.3 us); every pass
            // asks again only for the halves of 16 words somebody in the wave still misses one of (quarters of eight: four
            // round trips in a row for a full pass, slower), so the pass that finds the slowest workgroup's record is a
            // round trip of 16 loads, not of all 32
            uint32_t missing = 0xffffffffu;
            float r[32];
#pragma unroll
            for (int u = 0; u < 32; ++u)
                if (b0 + 8 * u >= numBlocks) { r[u] = 0.0f; missing &= ~(1u << u); }
            bool seen = false;
            for (uint32_t it = 0; it < spinLimit && !seen; ++it) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (__builtin_amdgcn_ballot_w64((missing >> (16 * h) & 0xffffu) != 0u) == 0ull) continue;
                    unsigned long long w[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        const int b = b0 + 8 * (16 * h + u);
                        w[u] = icp_load_word(&records[(size_t)(b < numBlocks ? b : b0) * kIcpStride + kk]);
                    }
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        const bool here = (int)(w[u] >> 32) == want && (missing >> (16 * h + u) & 1u);
                        r[16 * h + u] = here ? __builtin_bit_cast(float, (unsigned)w[u]) : r[16 * h + u];
                        missing = here ? missing & ~(1u << (16 * h + u)) : missing;
                    }
                }
                seen = __syncthreads_and(missing == 0u) != 0;
            }
            if (!seen) { timedOut = true; break; }
#pragma unroll
            for (int u = 0; u < 32; ++u) s += r[u];
        }
        VH_ICP_STAMP(3);
        sm[part][k] = s;
        __syncthreads();
        if (threadIdx.x < kIcpStride) {
            float v = 0.0f;
#pragma unroll
            for (int q8 = 0; q8 < 8; ++q8) v += sm[q8][threadIdx.x];
            total[threadIdx.x] = threadIdx.x < kIcpTerms ? v : 0.0f;
        }
        __syncthreads();
        VH_ICP_STAMP(4);
        if (threadIdx.x == 0) {
            double T[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) T[i] = sT[i];
            int done = 0, singular = 0;
            if (timedOut) { done = 1; __hip_atomic_store(&state->timeout, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
            else if (total[27] == 0.0f) done = 1;                              // CameraTracking.cpp:52
            else if (!icp_step_from_sums(total, T)) done = singular = 1;       // (T untouched)
#pragma unroll
            for (int i = 0; i < 12; ++i) { sT[i] = T[i]; sDelta[i] = (float)T[i]; }
            go = !done;
            if (done || finalRound) {                       // what the host reads when the launch has ended: `state` is host
                                                            // memory here (pinned, mapped) -- no copy behind the launch
#pragma unroll
                for (int i = 0; i < 16; ++i) { state->T[i] = T[i]; state->delta[i] = (float)T[i]; }
#pragma unroll
                for (int i = 0; i < kIcpStride; ++i) state->sums[i] = total[i];
                state->rounds = timedOut ? round : round + 1;
                state->done = done;
                state->singular = singular;
            }
        }
        __syncthreads();
        if (!finalRound && threadIdx.x < kIcpPubCopies * kIcpPubStride && (threadIdx.x & (kIcpPubStride - 1)) < 12)
            icp_store_word(pub + threadIdx.x, icp_word(sDelta[threadIdx.x & (kIcpPubStride - 1)], go ? want : -want));
        VH_ICP_STAMP(5);
        if (!go || finalRound) return;
    }
#undef VH_ICP_STAMP
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
