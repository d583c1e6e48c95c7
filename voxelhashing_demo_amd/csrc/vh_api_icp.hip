// vh_api_icp.hip -- C-ABI of the frame-to-frame ICP (include/voxelhash.h "camera tracking").
// Included at the end of vh_api.hip (same translation unit: shares fail(), VH_HIP, DeviceGuard).
// Host side of Solver::BuildLinearSystem (Solver.cpp:48-111), SE3Exp / SE3Log (SE3.cpp:4-22) and
// CameraTracking::Align (CameraTracking.cpp:27-69); the device side is vh_icp.hip.

struct vh_icp {
    int device = 0;
    int width = 0, height = 0;
    hipStream_t stream = nullptr;
    float *partials = nullptr;     // [blocks][32]
    float *sums = nullptr;         // [32] device
    float *hostSums = nullptr;     // [32] pinned
    int blocks = 0;
};

extern "C" int vh_icp_create(int32_t width, int32_t height, int32_t device, vh_icp **out)
{
    if (!out || width <= 0 || height <= 0 || (uint64_t)width * height > (1u << 24))
        return fail(VH_ERR_INVALID_ARGUMENT, "bad argument");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(VH_ERR_NO_DEVICE, "hipGetDeviceCount");
    int dev = device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return fail(VH_ERR_NO_DEVICE, "hipGetDevice");
    if (dev >= ndev) return fail(VH_ERR_INVALID_ARGUMENT, "device ordinal out of range");
    DeviceGuard guard(dev);
    if (!guard.ok) return fail(VH_ERR_NO_DEVICE, "hipSetDevice");
    vh_icp *p = new vh_icp();
    p->device = dev;
    p->width = width;
    p->height = height;
    p->blocks = grid_for((size_t)width * height, 256);
    hipError_t e = hipMalloc((void **)&p->partials, sizeof(float) * kIcpStride * (size_t)p->blocks);
    if (e == hipSuccess) e = hipMalloc((void **)&p->sums, sizeof(float) * kIcpStride);
    if (e == hipSuccess) e = hipHostMalloc((void **)&p->hostSums, sizeof(float) * kIcpStride, hipHostMallocDefault);
    if (e != hipSuccess) {
        if (p->partials) (void)hipFree(p->partials);
        if (p->sums) (void)hipFree(p->sums);
        delete p;
        return fail(e == hipErrorOutOfMemory ? VH_ERR_OUT_OF_MEMORY : VH_ERR_HIP, "icp workspace", e);
    }
    *out = p;
    return VH_OK;
}

extern "C" int vh_icp_destroy(vh_icp *p)
{
    if (!p) return VH_OK;
    DeviceGuard guard(p->device);
    (void)hipStreamSynchronize(p->stream);
    (void)hipFree(p->partials);
    (void)hipFree(p->sums);
    (void)hipHostFree(p->hostSums);
    delete p;
    return VH_OK;
}

extern "C" int vh_icp_set_stream(vh_icp *p, void *stream)
{
    if (!p) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    p->stream = (hipStream_t)stream;
    return VH_OK;
}

static int icp_launch(vh_icp *p, const vh_float4 *d_input, const vh_float4 *d_target, const vh_float4 *d_normals,
                      const float delta[16], const float K[9], float dist_thres, int flags, vh_float4 *d_corres,
                      vh_float4 *d_corres_normals, float *d_residuals, vh_icp_system *out)
{
    IcpParams ip;
    std::memcpy(ip.delta, delta, sizeof ip.delta);
    std::memcpy(ip.K, K, sizeof ip.K);
    ip.distThres = dist_thres;
    ip.width = p->width;
    ip.height = p->height;
    ip.flags = flags;
    icp_accumulate_kernel<<<p->blocks, 256, 0, p->stream>>>(
        ip, reinterpret_cast<const float4 *>(d_input), reinterpret_cast<const float4 *>(d_target),
        reinterpret_cast<const float4 *>(d_normals), p->partials, reinterpret_cast<float4 *>(d_corres),
        reinterpret_cast<float4 *>(d_corres_normals), d_residuals);
    icp_finalize_kernel<<<1, 256, 0, p->stream>>>(p->partials, p->blocks, p->sums);
    VH_HIP(hipGetLastError());
    VH_HIP(hipMemcpyAsync(p->hostSums, p->sums, sizeof(float) * kIcpStride, hipMemcpyDeviceToHost, p->stream));
    VH_HIP(hipStreamSynchronize(p->stream));
    int k = 0;
    for (int a = 0; a < 6; ++a)
        for (int b = a; b < 6; ++b) {
            out->JTJ[6 * a + b] = out->JTJ[6 * b + a] = (double)p->hostSums[k];
            ++k;
        }
    for (int a = 0; a < 6; ++a) out->JTr[a] = (double)p->hostSums[21 + a];
    out->error = (double)p->hostSums[27];
    out->count = (uint32_t)p->hostSums[28];
    return VH_OK;
}

extern "C" int vh_icp_build_system(vh_icp *p, const vh_float4 *d_input, const vh_float4 *d_target,
                                   const vh_float4 *d_target_normals, const float delta[16], const float K[9],
                                   float dist_thres, int32_t flags, vh_icp_system *out)
{
    if (!p || !d_input || !d_target || !d_target_normals || !delta || !K || !out)
        return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    DeviceGuard guard(p->device);
    return icp_launch(p, d_input, d_target, d_target_normals, delta, K, dist_thres, flags, nullptr, nullptr, nullptr, out);
}

extern "C" int vh_icp_correspondences(vh_icp *p, const vh_float4 *d_input, const vh_float4 *d_target,
                                      const vh_float4 *d_target_normals, const float delta[16], const float K[9],
                                      float dist_thres, int32_t flags, vh_float4 *d_corres,
                                      vh_float4 *d_corres_normals, float *d_residuals, vh_icp_system *out)
{
    if (!p || !d_input || !d_target || !d_target_normals || !delta || !K || !out || !d_corres || !d_corres_normals ||
        !d_residuals)
        return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    DeviceGuard guard(p->device);
    return icp_launch(p, d_input, d_target, d_target_normals, delta, K, dist_thres, flags, d_corres, d_corres_normals,
                      d_residuals, out);
}

// ---- SE3 (SE3.cpp:4-22): twist = (v, w), M = [[0,-w2,w1,v0],[w2,0,-w0,v1],[-w1,w0,0,v2],0]; the
// reference evaluates M.exp() / T.log() with Eigen's generic matrix functions, these are the
// closed forms of the same maps ----
static void skew_terms(const double w[3], double Kx[9], double K2[9])
{
    const double k[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
    std::memcpy(Kx, k, sizeof k);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) K2[3 * i + j] = k[3 * i] * k[j] + k[3 * i + 1] * k[3 + j] + k[3 * i + 2] * k[6 + j];
}

extern "C" void vh_se3_exp(const double twist[6], double T[16])
{
    const double *v = twist, *w = twist + 3;
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], th = std::sqrt(th2);
    double A, B, Cc;
    if (th < 1e-5) { A = 1.0 - th2 / 6.0; B = 0.5 - th2 / 24.0; Cc = 1.0 / 6.0 - th2 / 120.0; }
    else { A = std::sin(th) / th; B = (1.0 - std::cos(th)) / th2; Cc = (th - std::sin(th)) / (th2 * th); }
    double Kx[9], K2[9];
    skew_terms(w, Kx, K2);
    std::memset(T, 0, 16 * sizeof(double));
    for (int i = 0; i < 3; ++i) {
        double t = 0.0;
        for (int j = 0; j < 3; ++j) {
            const double I = (i == j) ? 1.0 : 0.0;
            T[4 * i + j] = I + A * Kx[3 * i + j] + B * K2[3 * i + j];
            t += (I + B * Kx[3 * i + j] + Cc * K2[3 * i + j]) * v[j];
        }
        T[4 * i + 3] = t;
    }
    T[15] = 1.0;
}

extern "C" void vh_se3_log(const double T[16], double twist[6])
{
    double c = 0.5 * (T[0] + T[5] + T[10] - 1.0);
    c = std::min(1.0, std::max(-1.0, c));
    const double th = std::acos(c), th2 = th * th;
    const double r[3] = {T[9] - T[6], T[2] - T[8], T[4] - T[1]};
    const double f = (th < 1e-5) ? 0.5 + th2 / 12.0 : th / (2.0 * std::sin(th));
    const double w[3] = {f * r[0], f * r[1], f * r[2]};
    const double D = (th < 1e-5) ? 1.0 / 12.0 + th2 / 720.0
                                 : (1.0 - th * std::sin(th) / (2.0 * (1.0 - std::cos(th)))) / th2;
    double Kx[9], K2[9];
    skew_terms(w, Kx, K2);
    for (int i = 0; i < 3; ++i) {
        double s = 0.0;
        for (int j = 0; j < 3; ++j) s += (((i == j) ? 1.0 : 0.0) - 0.5 * Kx[3 * i + j] + D * K2[3 * i + j]) * T[4 * j + 3];
        twist[i] = s;
    }
    twist[3] = w[0]; twist[4] = w[1]; twist[5] = w[2];
}

// update = -(JTJ^-1 JTr); estimate = log(exp(update) exp(estimate))   (Solver.cpp:104-106)
extern "C" int vh_icp_solve(const vh_icp_system *sys, double estimate[6])
{
    if (!sys || !estimate) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    double L[36] = {0}, y[6], x[6];
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = sys->JTJ[6 * i + j];
            for (int k = 0; k < j; ++k) s -= L[6 * i + k] * L[6 * j + k];
            if (i == j) {
                if (!(s > 0.0)) return fail(VH_ERR_SINGULAR, "J^T J is not positive definite");
                L[6 * i + i] = std::sqrt(s);
            } else {
                L[6 * i + j] = s / L[6 * j + j];
            }
        }
    for (int i = 0; i < 6; ++i) {
        double s = -sys->JTr[i];
        for (int k = 0; k < i; ++k) s -= L[6 * i + k] * y[k];
        y[i] = s / L[6 * i + i];
    }
    for (int i = 5; i >= 0; --i) {
        double s = y[i];
        for (int k = i + 1; k < 6; ++k) s -= L[6 * k + i] * x[k];
        x[i] = s / L[6 * i + i];
    }
    double A[16], B[16], M[16];
    vh_se3_exp(x, A);
    vh_se3_exp(estimate, B);
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0.0;
            for (int k = 0; k < 4; ++k) s += A[4 * i + k] * B[4 * k + j];
            M[4 * i + j] = s;
        }
    vh_se3_log(M, estimate);
    return VH_OK;
}

// CameraTracking::Align, CameraTracking.cpp:27-69
extern "C" int vh_icp_align(vh_icp *p, const vh_float4 *d_input, const vh_float4 *d_target,
                            const vh_float4 *d_target_normals, const float K[9], float dist_thres, int32_t max_iters,
                            int32_t flags, float delta[16], vh_icp_system *last, int32_t *iterations)
{
    if (!p || !d_input || !d_target || !d_target_normals || !K || !delta)
        return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    DeviceGuard guard(p->device);
    double T[16], est[6];
    for (int i = 0; i < 16; ++i) T[i] = (double)delta[i];
    vh_se3_log(T, est);
    vh_icp_system sys{};
    int it = 0;
    for (; it < max_iters; ++it) {
        float d32[16];
        vh_se3_exp(est, T);
        for (int i = 0; i < 16; ++i) d32[i] = (float)T[i];
        const int rc = icp_launch(p, d_input, d_target, d_target_normals, d32, K, dist_thres, flags, nullptr, nullptr,
                                  nullptr, &sys);
        if (rc != VH_OK) return rc;
        if (sys.error == 0.0) break;                                 // :52
        if (vh_icp_solve(&sys, est) != VH_OK) break;
    }
    vh_se3_exp(est, T);
    for (int i = 0; i < 16; ++i) delta[i] = (float)T[i];
    if (last) *last = sys;
    if (iterations) *iterations = it;
    return VH_OK;
}

extern "C" int vh_depth_to_maps(const float *d_depth, const float k_inv[9], int32_t width, int32_t height,
                                vh_float4 *d_positions, vh_float4 *d_normals, void *hip_stream)
{
    if (!d_depth || !k_inv || !d_positions || !d_normals || width <= 0 || height <= 0 ||
        (uint64_t)width * height > (1u << 24))
        return fail(VH_ERR_INVALID_ARGUMENT, "bad argument");
    Mat3 k;
    std::memcpy(k.m, k_inv, sizeof k.m);
    depth_to_maps_kernel<<<grid_for((size_t)width * height, 256), 256, 0, (hipStream_t)hip_stream>>>(
        d_depth, k, width, height, reinterpret_cast<float4 *>(d_positions), reinterpret_cast<float4 *>(d_normals));
    VH_HIP(hipGetLastError());
    return VH_OK;
}

// The reference's own name (CameraTrackingUtils.cu:187-215): 640x480, intrinsics from
// SetCameraIntrinsic, thresholds of common.h:12-13, synchronous; returns the summed residual.
// `deltaTransform` is a float4x4 passed by value in the reference; here a pointer to its 16
// row-major floats.
static vh_icp *g_icp = nullptr;

extern "C" float computeCorrespondences(const vh_float4 *d_input, const vh_float4 *d_target,
                                        const vh_float4 *d_targetNormals, vh_float4 *corres, vh_float4 *corresNormals,
                                        float *residual, const float *deltaTransform, int width, int height)
{
    int rc = VH_OK;
    if (g_icp && (g_icp->width != width || g_icp->height != height)) {
        vh_icp_destroy(g_icp);
        g_icp = nullptr;
    }
    if (!g_icp) rc = vh_icp_create(width, height, -1, &g_icp);
    vh_icp_system sys{};
    if (rc == VH_OK)
        rc = vh_icp_correspondences(g_icp, d_input, d_target, d_targetNormals, deltaTransform, g_k, 0.08f, 0, corres,
                                    corresNormals, residual, &sys);
    if (rc != VH_OK) {
        std::fprintf(stderr, "voxelhash: computeCorrespondences failed: %s (%s)\n", vh_error_string(rc), vh_last_error());
        std::exit(EXIT_FAILURE);
    }
    return (float)sys.error;
}
