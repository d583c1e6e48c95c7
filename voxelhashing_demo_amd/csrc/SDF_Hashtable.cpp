// SDF_Hashtable.cpp -- host facade over the C-ABI; mirrors SDF_Hashtable.cpp:11-40,60-90
// of the reference.  Errors follow the reference convention (checkCudaErrors,
// helper_cuda.h:966-977): message on stderr, then exit(EXIT_FAILURE).
#include "SDF_Hashtable.h"
#include "CameraTracking.h"

#include <cstdio>
#include <cstdlib>

static void check(int rc, const char *where)
{
    if (rc == VH_OK) return;
    std::fprintf(stderr, "SDF_Hashtable: %s failed: %s (%s)\n", where, vh_error_string(rc), vh_last_error());
    std::exit(EXIT_FAILURE);
}

SDF_Hashtable::SDF_Hashtable() : ctx_(nullptr), dist_(nullptr)
{
    vh_default_params(&h_hashtableParams);             // SDF_Hashtable.cpp:62-73
    vh_config cfg;
    cfg.params = h_hashtableParams;
    cfg.width = 640;                                   // common.h:17-18
    cfg.height = 480;
    cfg.semantics = VH_SEM_REFERENCE;
    cfg.device = -1;
    check(vh_create(&cfg, &ctx_), "deviceAllocate");   // :75-79
}

SDF_Hashtable::SDF_Hashtable(const HashTableParams &params, int width, int height, int semantics) : ctx_(nullptr), dist_(nullptr)
{
    h_hashtableParams = params;
    vh_config cfg;
    cfg.params = params;
    cfg.width = width;
    cfg.height = height;
    cfg.semantics = semantics;
    cfg.device = -1;
    check(vh_create(&cfg, &ctx_), "deviceAllocate");
}

// One rank of a table sharded over `world` GPUs (include/voxelhash_dist.h)
SDF_Hashtable::SDF_Hashtable(const HashTableParams &params, int width, int height, int semantics, int rank, int world, int batch,
                             const char uniqueId[VH_DIST_ID_BYTES], const float kInv[9], int device)
    : ctx_(nullptr), dist_(nullptr)
{
    h_hashtableParams = params;
    vh_dist_config cfg;
    cfg.table.params = params;
    cfg.table.width = width;
    cfg.table.height = height;
    cfg.table.semantics = semantics;
    cfg.table.device = device;
    cfg.rank = rank;
    cfg.world = world;
    cfg.batch = batch;
    cfg.key_capacity = 0;
    cfg.packet_format = VH_PACKET_U16;
    for (int i = 0; i < 9; ++i) cfg.k_inv[i] = kInv[i];
    check(vh_dist_create(&cfg, uniqueId, nullptr, &dist_), "vh_dist_create");
    ctx_ = vh_dist_shard(dist_);
}

void SDF_Hashtable::uniqueId(char id[VH_DIST_ID_BYTES]) { check(vh_dist_unique_id(id), "vh_dist_unique_id"); }
void SDF_Hashtable::loopbackId(char id[VH_DIST_ID_BYTES]) { check(vh_dist_loopback_id(id), "vh_dist_loopback_id"); }

void SDF_Hashtable::integrateExchange(const float *poses, const uint16_t *const *d_depth)
{
    check(dist_ ? vh_dist_step_batch(dist_, poses, reinterpret_cast<const void *const *>(d_depth)) : VH_ERR_INVALID_ARGUMENT,
          "integrateExchange");
}

SDF_Hashtable::~SDF_Hashtable()                          // :83-89
{
    if (dist_) vh_dist_destroy(dist_);                   // (owns the shard context)
    else vh_destroy(ctx_);
}

void SDF_Hashtable::integrate(const float4x4 &viewMat, const vh_float4 *verts, const vh_float4 *normals)
{
    // pose + inverse, mutex reset, allocBlocks, flattenIntoBuffer, integrateDepthMap
    // (:15-36) as one asynchronous submission
    check(vh_integrate(ctx_, viewMat.entries, verts, normals), "integrate");
}

void SDF_Hashtable::integrate(const float4x4 &viewMat, const uint16_t *d_depth, const float kInv[9])
{
    check(vh_integrate_depth(ctx_, viewMat.entries, d_depth, kInv), "integrate");
}

void SDF_Hashtable::raycast(const float4x4 &pose, float *d_depth_out, float zNear, float zFar)
{
    // sharded: through all shards, the record slots sized by the library until no view has holes (collective)
    if (dist_) check(vh_dist_raycast_auto(dist_, pose.entries, zNear, zFar, d_depth_out, nullptr, nullptr), "raycast");
    else check(vh_raycast(ctx_, pose.entries, zNear, zFar, d_depth_out), "raycast");
}

void SDF_Hashtable::raycast(const float4x4 &pose, float *d_depth_out, vh_float4 *d_normal_out, float zNear, float zFar)
{
    if (dist_) check(vh_dist_raycast_auto(dist_, pose.entries, zNear, zFar, d_depth_out, d_normal_out, nullptr), "raycast");
    else check(vh_raycast_normals(ctx_, pose.entries, zNear, zFar, d_depth_out, d_normal_out), "raycast");
}

void SDF_Hashtable::raycast(const float4x4 &pose, float *d_depth_out, vh_float4 *d_vertices_out, vh_float4 *d_normals_out,
                            float zNear, float zFar)
{
    // (a sharded table has no vertex-map form: the local shard alone would render a view full of holes)
    check(dist_ ? VH_ERR_INVALID_ARGUMENT : vh_raycast_maps(ctx_, pose.entries, zNear, zFar, d_depth_out, d_vertices_out, d_normals_out),
          dist_ ? "raycast with vertex maps is not available on a sharded table: use raycast(pose, depth, normals)" : "raycast");
}

void SDF_Hashtable::renderBlocks(const float4x4 &pose, float *d_front, float *d_back, float zNear, float zFar)
{
    check(vh_render_blocks(ctx_, pose.entries, zNear, zFar, d_front, d_back), "renderBlocks");
}

void SDF_Hashtable::garbageCollect(float sdfThreshold)
{
    check(vh_garbage_collect(ctx_, sdfThreshold), "garbageCollect");
}

int SDF_Hashtable::occupiedBlockCount()
{
    vh_counters c;
    check(vh_get_counters(ctx_, &c), "get_counters");
    h_hashtableParams.numOccupiedBlocks = (uint32_t)c.occupied;   // :32
    return c.occupied;
}

void SDF_Hashtable::setStream(void *s) { check(vh_set_stream(ctx_, s), "set_stream"); }
void SDF_Hashtable::setOption(const char *name, int value) { check(vh_set_option(ctx_, name, value), "set_option"); }
void SDF_Hashtable::setAllocBand(float bandMetres) { check(vh_set_alloc_band(ctx_, bandMetres), "set_alloc_band"); }
void SDF_Hashtable::flush() { check(dist_ ? vh_dist_flush(dist_) : vh_flush(ctx_), "flush"); }
void SDF_Hashtable::integrateBatch(int count, const float *poses, const vh_float4 *const *d_verts,
                                   const vh_float4 *const *d_normals)
{
    check(vh_integrate_batch(ctx_, count, poses, d_verts, d_normals), "integrate_batch");
}

// ---------------------------------------------------------------------------
// CameraTracking (CameraTracking.cpp:27-69,118-145)
// ---------------------------------------------------------------------------
CameraTracking::CameraTracking(int w, int h) : icp_(nullptr), width(w), height(h)
{
    deltaTransform.setIdentity();
    // common.h:7-10 scaled with the resolution
    const float sx = (float)w / 640.0f, sy = (float)h / 480.0f;
    const float K[9] = {517.3f * sx, 0, 318.6f * sx, 0, 516.5f * sy, 255.3f * sy, 0, 0, 1};
    setIntrinsic(K);
    check(vh_icp_create(w, h, -1, &icp_), "CameraTracking");
}

CameraTracking::~CameraTracking() { vh_icp_destroy(icp_); }

void CameraTracking::setIntrinsic(const float K[9])
{
    for (int i = 0; i < 9; ++i) K_[i] = K[i];
}

void CameraTracking::setStream(void *hipStream) { check(vh_icp_set_stream(icp_, hipStream), "setStream"); }

void CameraTracking::Align(vh_float4 *d_input, vh_float4 *, vh_float4 *d_target, vh_float4 *d_targetNormals,
                           const uint16_t *, const uint16_t *)
{
    vh_icp_system last;
    int rounds = 0;
    check(vh_icp_align(icp_, d_input, d_target, d_targetNormals, K_, distThres_, maxIters, flags_,
                       deltaTransform.entries, &last, &rounds), "Align");
    globalCorrespondenceError = (float)last.error;
}
