// vh_api_dropin.hip -- C-ABI, depth pre-processing and the reference's own names on a process-global context.
// Included by vh_api.hip (same translation unit: shares fail(), VH_HIP, DeviceGuard, launch()).

// ---------------------------------------------------------------------------
// depth pre-processing (CameraTrackingUtils.cu:115-120, 218-222)
// ---------------------------------------------------------------------------
extern "C" int vh_preprocess(const uint16_t *d_depth, const float k_inv[9], int32_t width, int32_t height,
                             vh_float4 *d_positions, vh_float4 *d_normals, void *hip_stream)
{
    VH_TRACE("vh_preprocess");
    if (!d_depth || !k_inv || !d_positions || !d_normals || width <= 0 || height <= 0 ||
        (uint64_t)width * height > (1u << 24))
        return fail(VH_ERR_INVALID_ARGUMENT, "bad argument");
    Mat3 k;
    std::memcpy(k.m, k_inv, sizeof k.m);
    preprocess_kernel<<<grid_for((size_t)width * height, 256), 256, 0, (hipStream_t)hip_stream>>>(
        d_depth, k, width, height, reinterpret_cast<float4 *>(d_positions), reinterpret_cast<float4 *>(d_normals));
    VH_HIP(hipGetLastError());
    return VH_OK;
}

static float g_k_inv[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
static float g_k[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};

extern "C" bool SetCameraIntrinsic(const float *intrinsic, const float *invIntrinsic)
{
    if (!invIntrinsic) return false;
    if (intrinsic) std::memcpy(g_k, intrinsic, sizeof g_k);          // K feeds computeCorrespondences
    std::memcpy(g_k_inv, invIntrinsic, sizeof g_k_inv);
    return true;
}

extern "C" void preProcess(vh_float4 *positions, vh_float4 *normals, const uint16_t *depth)
{
    // 640x480 and the default stream, like the reference (CameraTrackingUtils.cu:28-36,115-120)
    int rc = vh_preprocess(depth, g_k_inv, 640, 480, positions, normals, nullptr);
    if (rc == VH_OK && hipDeviceSynchronize() != hipSuccess) rc = VH_ERR_HIP;
    if (rc != VH_OK) {
        std::fprintf(stderr, "voxelhash: preProcess failed: %s (%s)\n", vh_error_string(rc), vh_last_error());
        std::exit(EXIT_FAILURE);
    }
}

// ---------------------------------------------------------------------------
// drop-in names (VoxelUtils.h:5-13) on a process-global context
// ---------------------------------------------------------------------------
static vh_context *g_default = nullptr;
static HashTableParams g_default_params;
static bool g_have_params = false;

[[noreturn]] static void die(const char *where, int rc)
{
    // checkCudaErrors convention, helper_cuda.h:966-977
    std::fprintf(stderr, "voxelhash: %s failed: %s (%s)\n", where, vh_error_string(rc), vh_last_error());
    std::exit(EXIT_FAILURE);
}

extern "C" vh_context *vh_default_context(void) { return g_default; }

extern "C" void updateConstantHashTableParams(const HashTableParams *params)
{
    // VoxelUtils.cu:87-91.  There is no __constant__ copy to refresh: kernels
    // receive the frame parameters by value.  The pose and the occupied count
    // are taken over.
    if (!params) die("updateConstantHashTableParams", VH_ERR_INVALID_ARGUMENT);
    g_default_params = *params;
    g_have_params = true;
    if (g_default) {
        std::memcpy(g_default->fp.T, params->global_transform, sizeof g_default->fp.T);
        std::memcpy(g_default->fp.Tinv, params->inv_global_transform, sizeof g_default->fp.Tinv);
        std::memcpy(g_default->params.global_transform, params->global_transform, sizeof g_default->fp.T);
        std::memcpy(g_default->params.inv_global_transform, params->inv_global_transform, sizeof g_default->fp.T);
        g_default->params.numOccupiedBlocks = params->numOccupiedBlocks;
    }
}

extern "C" void deviceAllocate(const HashTableParams *params)
{
    if (!params) die("deviceAllocate", VH_ERR_INVALID_ARGUMENT);
    if (g_default) { vh_destroy(g_default); g_default = nullptr; }
    vh_config cfg;
    cfg.params = *params;
    cfg.width = 640;          // common.h:17-18
    cfg.height = 480;
    cfg.semantics = VH_SEM_REFERENCE;
    cfg.device = -1;
    if (const char *s = std::getenv("VOXELHASH_SEMANTICS"))
        if (std::strcmp(s, "pinhole") == 0) cfg.semantics = VH_SEM_PINHOLE;
    int rc = vh_create(&cfg, &g_default);
    if (rc != VH_OK) die("deviceAllocate", rc);
}

extern "C" void deviceFree(void)
{
    if (g_default) { vh_destroy(g_default); g_default = nullptr; }
}

extern "C" void resetHashTableMutexes(const HashTableParams *params)
{
    (void)params;
    if (!g_default) die("resetHashTableMutexes", VH_ERR_NOT_INITIALISED);
    int rc = vh_reset_mutexes(g_default);
    if (rc != VH_OK) die("resetHashTableMutexes", rc);
}

extern "C" void allocBlocks(const vh_float4 *verts, const vh_float4 *normals)
{
    if (!g_default) die("allocBlocks", VH_ERR_NOT_INITIALISED);
    int rc = vh_alloc_blocks(g_default, verts, normals);
    if (rc == VH_OK) rc = vh_synchronize(g_default);     // the reference syncs after the launch (:715)
    if (rc != VH_OK) die("allocBlocks", rc);
}

extern "C" int flattenIntoBuffer(const HashTableParams *params)
{
    (void)params;
    if (!g_default) die("flattenIntoBuffer", VH_ERR_NOT_INITIALISED);
    int32_t n = 0;
    int rc = vh_flatten(g_default, &n);
    if (rc != VH_OK) die("flattenIntoBuffer", rc);
    return n;
}

extern "C" void calculateKinectProjectionMatrix(void)
{
    if (!g_default) die("calculateKinectProjectionMatrix", VH_ERR_NOT_INITIALISED);
    default_projection(g_default);                       // VoxelUtils.cu:224-231
}

extern "C" void integrateDepthMap(const HashTableParams *params, const vh_float4 *verts)
{
    if (!g_default) die("integrateDepthMap", VH_ERR_NOT_INITIALISED);
    if (params && params->numOccupiedBlocks == 0) return;          // :848
    int rc = vh_integrate_depth_map(g_default, verts);
    if (rc == VH_OK) rc = vh_synchronize(g_default);               // :850
    if (rc != VH_OK) die("integrateDepthMap", rc);
}
