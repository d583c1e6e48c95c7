// vh_api_icp.hip -- C-ABI of the frame-to-frame ICP (include/voxelhash.h "camera tracking").
// Included at the end of vh_api.hip (same translation unit: shares fail(), VH_HIP, DeviceGuard).
// Host side of Solver::BuildLinearSystem (Solver.cpp:48-111), SE3Exp / SE3Log (SE3.cpp:4-22) and
// CameraTracking::Align (CameraTracking.cpp:27-69); the device side is vh_icp.hip.

struct vh_icp {
    int device = 0;
    int width = 0, height = 0;
    hipStream_t stream = nullptr;
    float *partials = nullptr;     // [blocks][32]
    IcpState *state = nullptr;     // device-resident Align state
    IcpState *hostState = nullptr; // pinned copy (the one-launch Align writes its result here itself)
    IcpState *hostStateDev = nullptr;   // ... under this address
    int blocks = 0;                // grid of icp_round_kernel
    int alignBlocks = 0, alignSlots = -1;  // grid of icp_align_kernel and pixels per lane it keeps in registers (0: none, the points are
                                           // read again every round); -1: Align is a chain of one-launch rounds
    unsigned long long *records = nullptr, *pub = nullptr;   // one-launch Align: [alignBlocks][32] sums, [8][16] estimate, each word {value, seq}
    uint32_t spinLimit = 1u << 20;         // polls (~1 us each) before a workgroup of the one-launch Align gives up; VH_ICP_SPIN_LIMIT
    int seqBase = 0;                       // sequence numbers handed out so far (they only grow: nothing is reset between calls)
    unsigned long long *stamps = nullptr;  // diagnostics (VH_ICP_STAMPS=1): [round][8] time stamps of the one-launch Align
};

constexpr int kIcpAlignMaxSlots = 6;

static const void *icp_align_entry(int slots)
{
    switch (slots) {
#define VH_ICP_ALIGN(S) case S: return (const void *)icp_align_kernel<S>;
    VH_ICP_ALIGN(0) VH_ICP_ALIGN(1) VH_ICP_ALIGN(2) VH_ICP_ALIGN(3) VH_ICP_ALIGN(4) VH_ICP_ALIGN(5) VH_ICP_ALIGN(6)
#undef VH_ICP_ALIGN
    default: return nullptr;
    }
}

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
    const int maxBlocks = grid_for((size_t)width * height, kIcpThreads);
    // one workgroup per compute unit up to six pixels per lane (640x480: five), two beyond (1280x960, 20-round Align: 390 us
    // against 441 in one launch; as a chain of rounds 475 against 485, 455 with 384)
    p->blocks = std::min(maxBlocks, (size_t)width * height > (size_t)6 * 256 * kIcpThreads ? 512 : 256);
    if (const char *e = std::getenv("VH_ICP_BLOCKS")) p->blocks = std::max(1, std::min(maxBlocks, std::atoi(e)));   // tuning knob
    // Align in one launch on the round kernel's grid (the same partition, hence the same sums, as the chain of rounds): a lane
    // keeps its pixels' input points in registers when there are at most six of them (640x480: five), else reads them again
    // every round.  VH_ICP_PERSISTENT=0: the chain of one-launch rounds (A/B, tests).
    const size_t npix = (size_t)width * height;
    p->alignBlocks = p->blocks;
    p->alignSlots = (int)((npix + (size_t)p->alignBlocks * kIcpThreads - 1) / ((size_t)p->alignBlocks * kIcpThreads));
    if (p->alignSlots > kIcpAlignMaxSlots) p->alignSlots = 0;
    if (const char *e = std::getenv("VH_ICP_PERSISTENT")) if (std::atoi(e) == 0) p->alignSlots = -1;
    if (p->alignSlots >= 0) {
        // the one-launch Align waits on its own grid: every workgroup must be able to be resident at once (the kernel sits
        // at the edge of 256 registers per lane: one or two workgroups per compute unit depending on the compiler's mood; a
        // neighbour on another stream that holds compute units only delays the launch, and the waits are bounded)
        int perCu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, icp_align_entry(p->alignSlots), kIcpThreads, 0) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            p->alignBlocks > perCu * cus)
            p->alignSlots = -1;
    }
    if (const char *e = std::getenv("VH_ICP_SPIN_LIMIT")) p->spinLimit = (uint32_t)std::max(1, std::atoi(e));   // (tests: 1 = the time-out path)
    if (std::getenv("VH_ICP_STAMPS")) (void)hipMalloc((void **)&p->stamps, sizeof(unsigned long long) * (512 + 1024));
    hipError_t e = hipMalloc((void **)&p->partials, sizeof(float) * kIcpStride * (size_t)p->blocks);
    const size_t recordBytes = sizeof(unsigned long long) * kIcpStride * (size_t)p->alignBlocks;
    const size_t pubBytes = sizeof(unsigned long long) * kIcpPubCopies * kIcpPubStride;
    if (e == hipSuccess) e = hipMalloc((void **)&p->records, recordBytes);
    if (e == hipSuccess) e = hipMalloc((void **)&p->pub, pubBytes);
    if (e == hipSuccess) e = hipMemset(p->records, 0, recordBytes);
    if (e == hipSuccess) e = hipMemset(p->pub, 0, pubBytes);
    if (e == hipSuccess) e = hipMalloc((void **)&p->state, sizeof(IcpState));
    if (e == hipSuccess) e = hipHostMalloc((void **)&p->hostState, sizeof(IcpState), hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&p->hostStateDev, p->hostState, 0);
    if (e != hipSuccess) {
        if (p->partials) (void)hipFree(p->partials);
        if (p->records) (void)hipFree(p->records);
        if (p->pub) (void)hipFree(p->pub);
        if (p->state) (void)hipFree(p->state);
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
    (void)hipDeviceSynchronize();      // not p->stream: the caller's stream object may already be gone
    (void)hipFree(p->partials);
    (void)hipFree(p->records);
    (void)hipFree(p->pub);
    (void)hipFree(p->state);
    if (p->stamps) (void)hipFree(p->stamps);
    (void)hipHostFree(p->hostState);
    delete p;
    return VH_OK;
}

extern "C" int vh_icp_set_stream(vh_icp *p, void *stream)
{
    if (!p) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    p->stream = (hipStream_t)stream;
    return VH_OK;
}

static void system_from_sums(const float *sums, vh_icp_system *out)
{
    int k = 0;
    for (int a = 0; a < 6; ++a)
        for (int b = a; b < 6; ++b) {
            out->JTJ[6 * a + b] = out->JTJ[6 * b + a] = (double)sums[k];
            ++k;
        }
    for (int a = 0; a < 6; ++a) out->JTr[a] = (double)sums[21 + a];
    out->error = (double)sums[27];
    out->count = (uint32_t)sums[28];
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
    VH_HIP(hipMemsetAsync(p->state, 0, sizeof(IcpState), p->stream));
    if (d_corres)
        icp_round_kernel<true><<<p->blocks, kIcpThreads, 0, p->stream>>>(
            ip, reinterpret_cast<const float4 *>(d_input), reinterpret_cast<const float4 *>(d_target),
            reinterpret_cast<const float4 *>(d_normals), p->partials, reinterpret_cast<float4 *>(d_corres),
            reinterpret_cast<float4 *>(d_corres_normals), d_residuals, p->state, 0, 0);
    else
        icp_round_kernel<false><<<p->blocks, kIcpThreads, 0, p->stream>>>(
            ip, reinterpret_cast<const float4 *>(d_input), reinterpret_cast<const float4 *>(d_target),
            reinterpret_cast<const float4 *>(d_normals), p->partials, (float4 *)nullptr, (float4 *)nullptr,
            (float *)nullptr, p->state, 0, 0);
    VH_HIP(hipGetLastError());
    IcpState &hs = *p->hostState;
    VH_HIP(hipMemcpyAsync(&hs, p->state, sizeof hs, hipMemcpyDeviceToHost, p->stream));
    VH_HIP(hipStreamSynchronize(p->stream));
    system_from_sums(hs.sums, out);
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

// SE3 maps and the 6x6 solve: one implementation for host and device (vh_icp.hip)
extern "C" void vh_se3_exp(const double twist[6], double T[16]) { se3_exp_d(twist, T); }
extern "C" void vh_se3_log(const double T[16], double twist[6]) { se3_log_d(T, twist); }

extern "C" int vh_icp_solve(const vh_icp_system *sys, double estimate[6])
{
    if (!sys || !estimate) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    if (!icp_solve_d(sys->JTJ, sys->JTr, estimate)) return fail(VH_ERR_SINGULAR, "J^T J is not positive definite");
    return VH_OK;
}

// CameraTracking::Align, CameraTracking.cpp:27-69.  One launch runs all rounds (icp_align_kernel: the start value in the
// kernel arguments, the result written into the pinned host record, one synchronisation); images it does not take run a
// chain of one-launch rounds queued at once: round i's last workgroup solves the system on the device and leaves the new
// estimate where round i+1 reads it, rounds after a stop condition fall through, one copy each way.
extern "C" int vh_icp_align(vh_icp *p, const vh_float4 *d_input, const vh_float4 *d_target,
                            const vh_float4 *d_target_normals, const float K[9], float dist_thres, int32_t max_iters,
                            int32_t flags, float delta[16], vh_icp_system *last, int32_t *iterations)
{
    VH_TRACE("vh_icp_align");
    if (!p || !d_input || !d_target || !d_target_normals || !K || !delta || max_iters < 0)
        return fail(VH_ERR_INVALID_ARGUMENT, "bad argument");
    DeviceGuard guard(p->device);
    const auto hostT0 = std::chrono::steady_clock::now();
    IcpState &hs = *p->hostState;
    std::memset(&hs, 0, sizeof hs);
    // the start value goes through log / exp once, like the reference's estimate (a rigid-body
    // matrix comes back unchanged up to rounding, anything else is projected onto SE3)
    double T[16], est[6];
    for (int i = 0; i < 16; ++i) T[i] = (double)delta[i];
    se3_log_d(T, est);
    se3_exp_d(est, T);
    for (int i = 0; i < 16; ++i) { hs.T[i] = T[i]; hs.delta[i] = (float)T[i]; }
    const bool oneLaunch = p->alignSlots >= 0 && max_iters > 0;
    // (the chain of rounds keeps its state in device memory; the one-launch Align gets the start by value and writes the
    // result straight into the pinned host record: no copy command in front of the launch or behind it)
    if (!oneLaunch) VH_HIP(hipMemcpyAsync(p->state, &hs, sizeof hs, hipMemcpyHostToDevice, p->stream));
    IcpParams ip;
    std::memset(ip.delta, 0, sizeof ip.delta);
    std::memcpy(ip.K, K, sizeof ip.K);
    ip.distThres = dist_thres;
    ip.width = p->width;
    ip.height = p->height;
    ip.flags = flags;
    const float4 *in = reinterpret_cast<const float4 *>(d_input), *tg = reinterpret_cast<const float4 *>(d_target),
                 *tn = reinterpret_cast<const float4 *>(d_target_normals);
    if (oneLaunch) {
        // one launch for all rounds (icp_align_kernel)
        const uint32_t spinLimit = p->spinLimit;
        if (p->seqBase > (1 << 30)) {          // (after 5 * 10^7 calls: start the numbers again behind cleared words)
            VH_HIP(hipMemsetAsync(p->records, 0, sizeof(unsigned long long) * kIcpStride * (size_t)p->alignBlocks, p->stream));
            VH_HIP(hipMemsetAsync(p->pub, 0, sizeof(unsigned long long) * kIcpPubCopies * kIcpPubStride, p->stream));
            p->seqBase = 0;
        }
        const int seqBase = p->seqBase;
        p->seqBase += max_iters;
        IcpStart start;
        std::memcpy(start.T, T, sizeof start.T);
        void *args[] = {(void *)&ip, (void *)&in, (void *)&tg, (void *)&tn, (void *)&p->records, (void *)&p->pub, (void *)&start,
                        (void *)&p->hostStateDev, (void *)&max_iters, (void *)&seqBase, (void *)&spinLimit, (void *)&p->stamps};
        VH_HIP(hipLaunchKernel(icp_align_entry(p->alignSlots), dim3(p->alignBlocks), dim3(kIcpThreads), args, 0, p->stream));
    } else {
        for (int it = 0; it < max_iters; ++it)
            icp_round_kernel<false><<<p->blocks, kIcpThreads, 0, p->stream>>>(ip, in, tg, tn, p->partials, (float4 *)nullptr,
                                                                             (float4 *)nullptr, (float *)nullptr, p->state, 1, 1);
    }
    VH_HIP(hipGetLastError());
    if (!oneLaunch) VH_HIP(hipMemcpyAsync(&hs, p->state, sizeof hs, hipMemcpyDeviceToHost, p->stream));
    const auto hostT1 = std::chrono::steady_clock::now();
    VH_HIP(hipStreamSynchronize(p->stream));
    if (p->stamps) {          // diagnostics: host time of this call up to the last launch call returning, and inside the synchronisation
        const auto hostT2 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "icp host: queued after %.1f us, synchronised after %.1f us more\n",
                     std::chrono::duration<double, std::micro>(hostT1 - hostT0).count(),
                     std::chrono::duration<double, std::micro>(hostT2 - hostT1).count());
    }
    if (p->stamps && oneLaunch && max_iters <= 64) {          // diagnostics: microseconds since the round started
        unsigned long long st[64 * 8];
        VH_HIP(hipMemcpy(st, p->stamps, sizeof(unsigned long long) * 8 * max_iters, hipMemcpyDeviceToHost));
        for (int r = 0; r < max_iters; ++r) {
            std::fprintf(stderr, "icp stamps round %2d:", r);
            for (int i = 1; i < 8; ++i) std::fprintf(stderr, " [%d] %6.2f", i, ((double)st[8 * r + i] - (double)st[8 * r]) * 0.01);
            if (r + 1 < max_iters) std::fprintf(stderr, "  next round %6.2f", ((double)st[8 * r + 8] - (double)st[8 * r]) * 0.01);
            std::fprintf(stderr, "\n");
        }
    }
    if (p->stamps && oneLaunch && max_iters > 11) {          // round 10: when each workgroup had stored its record
        static unsigned long long st[512 + 1024];
        VH_HIP(hipMemcpy(st, p->stamps, sizeof st, hipMemcpyDeviceToHost));
        std::fprintf(stderr, "icp stamps round 10, record stored (us after workgroup 0 started the round), by workgroup:");
        for (int b = 0; b < p->alignBlocks; ++b) std::fprintf(stderr, "%s%5.2f", b % 16 ? " " : "\n  ", ((double)st[512 + b] - (double)st[80]) * 0.01);
        std::fprintf(stderr, "\n");
    }
    if (hs.timeout) return fail(VH_ERR_TIMEOUT, "vh_icp_align: a workgroup of the one-launch Align gave up waiting for a round's estimate");
    std::memcpy(delta, hs.delta, 16 * sizeof(float));
    if (last) system_from_sums(hs.sums, last);
    // rounds that solved = systems built, minus the one that hit a stop condition (the oracle's count)
    if (iterations) *iterations = hs.rounds - (hs.done ? 1 : 0);
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

// vh_raycast followed by vh_depth_to_maps with the K^-1 of the raycast intrinsics: depth, camera-frame
// vertex map and normal map of the model seen from `pose` (raycast(pose, d_depth_out,
// d_normal_out)).  The maps are what vh_icp_align takes as its target.
extern "C" int vh_raycast_maps(vh_context *c, const float pose[16], float t_min, float t_max, float *d_depth_out,
                               vh_float4 *d_vertices_out, vh_float4 *d_normals_out)
{
    if (!c || !d_depth_out || !d_vertices_out || !d_normals_out) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    const int rc = vh_raycast(c, pose, t_min, t_max, d_depth_out);
    if (rc != VH_OK) return rc;
    const float k_inv[9] = {1.0f / c->rc_fx, 0.0f, -c->rc_cx / c->rc_fx, 0.0f, 1.0f / c->rc_fy, -c->rc_cy / c->rc_fy,
                            0.0f, 0.0f, 1.0f};
    DeviceGuard guard(c->device);
    return vh_depth_to_maps(d_depth_out, k_inv, c->fp.width, c->fp.height, d_vertices_out, d_normals_out, c->stream);
}

// One frame of the closed loop in one call (frame order of Application.cpp:73-90; tracking.FusionLoop.step):
// preProcess -> Align against the model's maps -> pose <- pose . delta -> integrate at the new pose -> the model's maps
// from there for the next frame.  Everything on the context's stream; the one host synchronisation is vh_icp_align's.
extern "C" int vh_fusion_step(vh_context *c, vh_icp *p, const uint16_t *d_depth, const float k_inv[9], const float K[9],
                              float dist_thres, int32_t max_iters, int32_t flags, float t_min, float t_max,
                              vh_float4 *d_input_vertices, vh_float4 *d_input_normals, float *d_model_depth,
                              vh_float4 *d_model_vertices, vh_float4 *d_model_normals, double pose[16],
                              vh_icp_system *last, int32_t *iterations)
{
    if (!c || !p || !d_depth || !k_inv || !K || !d_input_vertices || !d_input_normals || !d_model_depth || !d_model_vertices ||
        !d_model_normals || !pose)
        return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    if (p->width != c->fp.width || p->height != c->fp.height || p->device != c->device || p->stream != c->stream)
        return fail(VH_ERR_INVALID_ARGUMENT, "vh_fusion_step: the tracker must have the table's image size, device and stream");
    int rc = vh_preprocess(d_depth, k_inv, p->width, p->height, d_input_vertices, d_input_normals, c->stream);
    if (rc != VH_OK) return rc;
    float delta[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    rc = vh_icp_align(p, d_input_vertices, d_model_vertices, d_model_normals, K, dist_thres, max_iters, flags, delta, last, iterations);
    if (rc != VH_OK) return rc;
    double next[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0.0;
            for (int k = 0; k < 4; ++k) s += pose[4 * i + k] * (double)delta[4 * k + j];
            next[4 * i + j] = s;
        }
    float p32[16];
    for (int i = 0; i < 16; ++i) { pose[i] = next[i]; p32[i] = (float)next[i]; }
    rc = vh_integrate_depth(c, p32, d_depth, k_inv);
    if (rc != VH_OK) return rc;
    return vh_raycast_maps(c, p32, t_min, t_max, d_model_depth, d_model_vertices, d_model_normals);
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
