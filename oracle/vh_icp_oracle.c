/* vh_icp_oracle.c -- CPU restatement of the reference's frame-to-frame point-to-plane ICP
 * (second half).  TEST INFRASTRUCTURE like vh_oracle.c: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * Reference pieces restated (the reference never calls them: tracker->Align is commented out,
 * Application.cpp:75):
 *   FindCorrespondences / computeCorrespondences   CameraTrackingUtils.cu:122-215
 *   CalculateJacobians / CalculateJacAndResKernel  Solver.cu:19-54
 *   Solver::BuildLinearSystem (cublasSgemv + cublasSsyrk + inverse)   Solver.cpp:48-111
 *   SE3Exp / SE3Log                                 SE3.cpp:4-22
 *   CameraTracking::Align                           CameraTracking.cpp:27-69
 * Parity status: "parity unpinned" (no fixtures in the reference, CUDA + cuBLAS + Eigen not
 * buildable here).  The per-pixel arithmetic is fp32 in the reference's operation order; the
 * sums (cuBLAS in the reference, order unspecified) are accumulated in double here, so the HIP
 * path's fp32 tree sums are compared with a tolerance. */
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "vh_oracle.h"

/* double -> int as cvt.rzi.s32.f64 does it: truncate, saturate, NaN -> 0 */
static int32_t d2i_rz(double x)
{
    if (x != x) return 0;
    if (x >= 2147483648.0) return INT32_MAX;
    if (x <= -2147483649.0) return INT32_MIN;
    return (int32_t)x;
}

/* Vertex / normal maps from a float depth image in metres (0 = no measurement): the
 * arithmetic of preProcess (CameraTrackingUtils.cu:50-113) without the /5000 of the uint16
 * path.  This is what turns a raycast depth image into an ICP target. */
void vho_depth_to_maps(const float *depth, const float k_inv[9], int W, int H, float *positions, float *normals)
{
    for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
        const float d = depth[(size_t)y * W + x];
        const float fx = (float)x, fy = (float)y;
        float *v = positions + 4 * ((size_t)y * W + x);
        v[0] = (k_inv[0] * fx + k_inv[1] * fy + k_inv[2] * 1.0f) * d;
        v[1] = (k_inv[3] * fx + k_inv[4] * fy + k_inv[5] * 1.0f) * d;
        v[2] = (k_inv[6] * fx + k_inv[7] * fy + k_inv[8] * 1.0f) * d;
        v[3] = 1.0f;
    }
    for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
        float *n = normals + 4 * ((size_t)y * W + x);
        n[0] = n[1] = n[2] = n[3] = 0.0f;
        if (!(x > 0 && x < W - 1 && y > 0 && y < H - 1)) continue;
        const float *CC = positions + 4 * ((size_t)y * W + x);
        const float *PC = positions + 4 * ((size_t)(y + 1) * W + x);
        const float *CP = positions + 4 * ((size_t)y * W + x + 1);
        const float *MC = positions + 4 * ((size_t)(y - 1) * W + x);
        const float *CM = positions + 4 * ((size_t)y * W + x - 1);
        if (!(CC[0] != 0 && PC[0] != 0 && CP[0] != 0 && MC[0] != 0 && CM[0] != 0)) continue;
        const float a[3] = { PC[0] - MC[0], PC[1] - MC[1], PC[2] - MC[2] };
        const float b[3] = { CP[0] - CM[0], CP[1] - CM[1], CP[2] - CM[2] };
        const float c[3] = { a[1]*b[2] - a[2]*b[1], a[2]*b[0] - a[0]*b[2], a[0]*b[1] - a[1]*b[0] };
        const float l = sqrtf(c[0]*c[0] + c[1]*c[1] + c[2]*c[2]);
        if (l > 0.0f) { n[0] = c[0] / l; n[1] = c[1] / l; n[2] = c[2] / l; }
    }
}

/* One pass of FindCorrespondences + CalculateJacAndResKernel + JTr = J^T r, JTJ = J^T J.
 *   source pixel idx is used when input[idx].z != 0                           (:148)
 *   q = delta * (p, 1); screen = (int)(K q / (K q).z + 0.5), the + 0.5 in double      (:124-129)
 *   kept when 0 < sx < W and 0 < sy < H (strict: column / row 0 never match)    (:157)
 *   d = dot(q - target[screen], normal[screen]); kept when d < dist_thres (signed, :170)
 *   row of J = [n, target x n], residual = d                                   (Solver.cu:27-35)
 * flags: VHO_ICP_ABS_DISTANCE: |d| < dist_thres; VHO_ICP_NEED_TARGET: target z != 0 and a
 * non-zero normal are required.  out: JTJ[36] row-major symmetric, JTr[6], error = sum d,
 * count = correspondences kept. */
static void build_system(const float *input, const float *target, const float *target_normals,
                         const float delta[16], const float K[9], float dist_thres, int W, int H, int flags,
                         double JTJ[36], double JTr[6], double *error, uint32_t *count,
                         float *corres, float *corres_normals, float *residuals)
{
    if (corres) {                                   /* computeCorrespondences clears the maps first, :198-200 */
        memset(corres, 0, (size_t)W * H * 4 * sizeof(float));
        memset(corres_normals, 0, (size_t)W * H * 4 * sizeof(float));
        memset(residuals, 0, (size_t)W * H * sizeof(float));
    }
    memset(JTJ, 0, 36 * sizeof(double));
    memset(JTr, 0, 6 * sizeof(double));
    double err = 0.0;
    uint32_t n = 0;
    for (int idx = 0; idx < W * H; ++idx) {
        const float *p = input + 4 * (size_t)idx;
        if (p[2] == 0.0f) continue;
        float q[3];
        for (int r = 0; r < 3; ++r)
            q[r] = delta[4*r+0] * p[0] + delta[4*r+1] * p[1] + delta[4*r+2] * p[2] + delta[4*r+3] * 1.0f;
        const float sx = K[0] * q[0] + K[1] * q[1] + K[2] * q[2];
        const float sy = K[3] * q[0] + K[4] * q[1] + K[5] * q[2];
        const float sz = K[6] * q[0] + K[7] * q[1] + K[8] * q[2];
        const int32_t u = d2i_rz((double)(sx / sz) + 0.5);
        const int32_t v = d2i_rz((double)(sy / sz) + 0.5);
        if (!(u > 0 && v > 0 && u < W && v < H)) continue;
        const float *t = target + 4 * ((size_t)v * W + u);
        const float *nn = target_normals + 4 * ((size_t)v * W + u);
        if ((flags & VHO_ICP_NEED_TARGET) && (t[2] == 0.0f || (nn[0] == 0.0f && nn[1] == 0.0f && nn[2] == 0.0f)))
            continue;
        const float dx = q[0] - t[0], dy = q[1] - t[1], dz = q[2] - t[2];
        const float d = dx * nn[0] + dy * nn[1] + dz * nn[2];
        const int keep = (flags & VHO_ICP_ABS_DISTANCE) ? (fabsf(d) < dist_thres) : (d < dist_thres);
        if (!keep) continue;
        const float J[6] = { nn[0], nn[1], nn[2],
                             t[1] * nn[2] - t[2] * nn[1], t[2] * nn[0] - t[0] * nn[2], t[0] * nn[1] - t[1] * nn[0] };
        for (int a = 0; a < 6; ++a) {
            JTr[a] += (double)(J[a] * d);
            for (int b = 0; b < 6; ++b) JTJ[6*a+b] += (double)(J[a] * J[b]);
        }
        err += (double)d;
        ++n;
        if (corres) {                               /* :176-178 */
            memcpy(corres + 4 * (size_t)idx, t, 4 * sizeof(float));
            memcpy(corres_normals + 4 * (size_t)idx, nn, 4 * sizeof(float));
            residuals[idx] = d;
        }
    }
    *error = err;
    *count = n;
}

void vho_icp_build_system(const float *input, const float *target, const float *target_normals,
                          const float delta[16], const float K[9], float dist_thres, int W, int H, int flags,
                          double JTJ[36], double JTr[6], double *error, uint32_t *count)
{
    build_system(input, target, target_normals, delta, K, dist_thres, W, H, flags, JTJ, JTr, error, count, NULL, NULL,
                 NULL);
}

/* computeCorrespondences (CameraTrackingUtils.cu:187-215): the same pass, also filling the target
 * point / target normal / residual of every kept source pixel (zero elsewhere); returns sum d. */
double vho_icp_correspondences(const float *input, const float *target, const float *target_normals,
                               const float delta[16], const float K[9], float dist_thres, int W, int H, int flags,
                               float *corres, float *corres_normals, float *residuals, uint32_t *count)
{
    double JTJ[36], JTr[6], err;
    build_system(input, target, target_normals, delta, K, dist_thres, W, H, flags, JTJ, JTr, &err, count, corres,
                 corres_normals, residuals);
    return err;
}

/* ---- SE3 (SE3.cpp:4-22; twist = (v, w): M = [[0,-w2,w1,v0],[w2,0,-w0,v1],[-w1,w0,0,v2],0]) ---- */
static void mat4_mul(const double a[16], const double b[16], double out[16])
{
    double r[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0.0;
            for (int k = 0; k < 4; ++k) s += a[4*i+k] * b[4*k+j];
            r[4*i+j] = s;
        }
    memcpy(out, r, sizeof r);
}

void vho_se3_exp(const double twist[6], double T[16])
{
    const double *v = twist, *w = twist + 3;
    const double th2 = w[0]*w[0] + w[1]*w[1] + w[2]*w[2], th = sqrt(th2);
    double A, B, C;                       /* sin(t)/t, (1-cos t)/t^2, (t-sin t)/t^3 */
    if (th < 1e-5) { A = 1.0 - th2 / 6.0; B = 0.5 - th2 / 24.0; C = 1.0 / 6.0 - th2 / 120.0; }
    else { A = sin(th) / th; B = (1.0 - cos(th)) / th2; C = (th - sin(th)) / (th2 * th); }
    const double Kx[9] = { 0, -w[2], w[1],  w[2], 0, -w[0],  -w[1], w[0], 0 };
    double K2[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) s += Kx[3*i+k] * Kx[3*k+j];
            K2[3*i+j] = s;
        }
    memset(T, 0, 16 * sizeof(double));
    for (int i = 0; i < 3; ++i) {
        double t = 0.0;
        for (int j = 0; j < 3; ++j) {
            const double I = (i == j) ? 1.0 : 0.0;
            T[4*i+j] = I + A * Kx[3*i+j] + B * K2[3*i+j];
            t += (I + B * Kx[3*i+j] + C * K2[3*i+j]) * v[j];
        }
        T[4*i+3] = t;
    }
    T[15] = 1.0;
}

void vho_se3_log(const double T[16], double twist[6])
{
    const double tr = T[0] + T[5] + T[10];
    double c = 0.5 * (tr - 1.0);
    if (c > 1.0) c = 1.0;
    if (c < -1.0) c = -1.0;
    const double th = acos(c);
    double w[3];
    const double r[3] = { T[9] - T[6], T[2] - T[8], T[4] - T[1] };       /* (R - R^T) vee */
    const double f = (th < 1e-5) ? 0.5 + th * th / 12.0 : th / (2.0 * sin(th));
    for (int i = 0; i < 3; ++i) w[i] = f * r[i];
    const double th2 = th * th;
    const double D = (th < 1e-5) ? 1.0 / 12.0 + th2 / 720.0 : (1.0 - th * sin(th) / (2.0 * (1.0 - cos(th)))) / th2;
    const double Kx[9] = { 0, -w[2], w[1],  w[2], 0, -w[0],  -w[1], w[0], 0 };
    for (int i = 0; i < 3; ++i) {
        double s = 0.0;
        for (int j = 0; j < 3; ++j) {
            double k2 = 0.0;
            for (int k = 0; k < 3; ++k) k2 += Kx[3*i+k] * Kx[3*k+j];
            s += (((i == j) ? 1.0 : 0.0) - 0.5 * Kx[3*i+j] + D * k2) * T[4*j+3];
        }
        twist[i] = s;
    }
    twist[3] = w[0]; twist[4] = w[1]; twist[5] = w[2];
}

/* update = -(JTJ^-1 JTr); estimate = log(exp(update) exp(estimate))  (Solver.cpp:104-106).
 * The 6x6 system is solved by Cholesky; returns 0 when JTJ is not positive definite (the
 * reference would produce inf / nan through JTJ.inverse()). */
int vho_icp_solve(const double JTJ[36], const double JTr[6], double estimate[6])
{
    double L[36] = {0}, y[6], x[6];
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = JTJ[6*i+j];
            for (int k = 0; k < j; ++k) s -= L[6*i+k] * L[6*j+k];
            if (i == j) {
                if (!(s > 0.0)) return 0;
                L[6*i+i] = sqrt(s);
            } else L[6*i+j] = s / L[6*j+j];
        }
    for (int i = 0; i < 6; ++i) {
        double s = -JTr[i];
        for (int k = 0; k < i; ++k) s -= L[6*i+k] * y[k];
        y[i] = s / L[6*i+i];
    }
    for (int i = 5; i >= 0; --i) {
        double s = y[i];
        for (int k = i + 1; k < 6; ++k) s -= L[6*k+i] * x[k];
        x[i] = s / L[6*i+i];
    }
    double A[16], B[16], Cm[16];
    vho_se3_exp(x, A);
    vho_se3_exp(estimate, B);
    mat4_mul(A, B, Cm);
    vho_se3_log(Cm, estimate);
    return 1;
}

/* CameraTracking::Align (CameraTracking.cpp:27-69): up to max_iters rounds of
 * correspondences -> system -> solve; stops when the summed residual is exactly 0 (:52) or the
 * system is singular.  delta (row-major, maps input points into the target's camera frame) is
 * both the start value and the result; returns the rounds executed. */
int vho_icp_align(const float *input, const float *target, const float *target_normals, const float K[9],
                  float dist_thres, int W, int H, int max_iters, int flags, float delta[16], double *final_error,
                  uint32_t *final_count)
{
    double T[16], est[6];
    for (int i = 0; i < 16; ++i) T[i] = (double)delta[i];
    vho_se3_log(T, est);
    int it = 0;
    double err = 0.0;
    uint32_t cnt = 0;
    for (; it < max_iters; ++it) {
        double JTJ[36], JTr[6];
        float d32[16];
        vho_se3_exp(est, T);
        for (int i = 0; i < 16; ++i) d32[i] = (float)T[i];
        vho_icp_build_system(input, target, target_normals, d32, K, dist_thres, W, H, flags, JTJ, JTr, &err, &cnt);
        if (err == 0.0) break;
        if (!vho_icp_solve(JTJ, JTr, est)) break;
    }
    vho_se3_exp(est, T);
    for (int i = 0; i < 16; ++i) delta[i] = (float)T[i];
    if (final_error) *final_error = err;
    if (final_count) *final_count = cnt;
    return it;
}
