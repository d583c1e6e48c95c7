// SDF_Hashtable.cpp -- host facade over the C-ABI; mirrors SDF_Hashtable.cpp:11-40,60-90
// of the reference.  Errors follow the reference convention (checkCudaErrors,
// helper_cuda.h:966-977): message on stderr, then exit(EXIT_FAILURE).
#include "SDF_Hashtable.h"

#include <cstdio>
#include <cstdlib>

static void check(int rc, const char *where)
{
    if (rc == VH_OK) return;
    std::fprintf(stderr, "SDF_Hashtable: %s failed: %s (%s)\n", where, vh_error_string(rc), vh_last_error());
    std::exit(EXIT_FAILURE);
}

SDF_Hashtable::SDF_Hashtable() : ctx_(nullptr)
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

SDF_Hashtable::SDF_Hashtable(const HashTableParams &params, int width, int height, int semantics) : ctx_(nullptr)
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

SDF_Hashtable::~SDF_Hashtable() { vh_destroy(ctx_); }   // :83-89

void SDF_Hashtable::integrate(const float4x4 &viewMat, const vh_float4 *verts, const vh_float4 *normals)
{
    // pose + inverse, mutex reset, allocBlocks, flattenIntoBuffer, integrateDepthMap
    // (:15-36) as one asynchronous submission
    check(vh_integrate(ctx_, viewMat.entries, verts, normals), "integrate");
}

void SDF_Hashtable::raycast(const float4x4 &pose, float *d_depth_out, float zNear, float zFar)
{
    check(vh_raycast(ctx_, pose.entries, zNear, zFar, d_depth_out), "raycast");
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
