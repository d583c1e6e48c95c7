/*
 * SDF_Hashtable.h -- C++ host facade with the reference's class interface
 * (SDF_Hashtable.h:24-42 / SDF_Hashtable.cpp) on top of the C-ABI in
 * voxelhash.h.  Source-compatible for the hot path:
 *
 *     SDF_Hashtable table;                       // common.h defaults
 *     table.integrate(pose, d_verts, d_normals); // SDF_Hashtable.cpp:11-40
 *
 * The GL interop members (registerGLtoCUDA / unmapCUDApointers,
 * SDF_Hashtable.cpp:42-58) are kept as no-ops: the compact table, its counter
 * and the SDF volume are library-owned device buffers.  raycast() stands in
 * for SDFRenderer::render(const glm::mat4&) (SDFRenderer.h:38).
 */
#ifndef SDF_HASHTABLE_H
#define SDF_HASHTABLE_H

#include <cstdint>

#include "voxelhash.h"
#include "voxelhash_dist.h"

/* row-major 4x4, the only part of cuda_SimpleMatrixUtil.h:800-1100 the path uses */
struct float4x4 {
    float entries[16];
    float4x4() {}
    explicit float4x4(const float values[16]) { for (int i = 0; i < 16; ++i) entries[i] = values[i]; }
    void setIdentity() { for (int i = 0; i < 16; ++i) entries[i] = (i % 5 == 0) ? 1.0f : 0.0f; }
    float &operator()(int r, int c) { return entries[4 * r + c]; }
    float operator()(int r, int c) const { return entries[4 * r + c]; }
};

class SDFRenderer;   /* not part of this build; kept so signatures compile */

class SDF_Hashtable {
    vh_context *ctx_;
    vh_dist *dist_;                                    /* multi-GPU constructor: this rank of the sharded table (owns ctx_) */
    HashTableParams h_hashtableParams;

public:
    SDF_Hashtable();                                   /* common.h:39-50, 640x480, REFERENCE semantics */
    SDF_Hashtable(const HashTableParams &params, int width, int height, int semantics);
    /* One rank of ONE logical table sharded by bucket range over `world` GPUs, one camera per rank (voxelhash_dist.h):
     * params.numBuckets = all buckets, params.numVoxelBlocks = per rank; uniqueId: the 128 bytes of
     * SDF_Hashtable::uniqueId() drawn by one rank and handed to all (any out-of-band channel); `batch` frames per
     * camera travel in one exchange; kInv: K^-1 of the cameras (the frames are uint16 sensor images).  The frames enter
     * through integrateExchange(); raycast() renders this rank's view through the whole table. */
    SDF_Hashtable(const HashTableParams &params, int width, int height, int semantics, int rank, int world, int batch,
                  const char uniqueId[VH_DIST_ID_BYTES], const float kInv[9], int device = -1);
    static void uniqueId(char id[VH_DIST_ID_BYTES]);
    /* the id of an in-process loop-back group (vh_dist_loopback_id): `world` SDF_Hashtable ranks of ONE process, one host
     * thread per rank, exchange by device copies instead of RCCL -- the N > 1 host on a single-GPU box */
    static void loopbackId(char id[VH_DIST_ID_BYTES]);
    /* `batch` frames of this rank's camera (poses: batch*16 row-major floats; d_depth: batch device pointers): queues
     * this exchange and applies an earlier one (vh_dist_step_batch: the one before the previous by default); flush() completes
     * what is in flight.  Collective over the ranks. */
    void integrateExchange(const float *poses, const uint16_t *const *d_depth);
    bool sharded() const { return dist_ != nullptr; }
    ~SDF_Hashtable();
    SDF_Hashtable(const SDF_Hashtable &) = delete;
    SDF_Hashtable &operator=(const SDF_Hashtable &) = delete;

    void integrate(const float4x4 &deltaT, const vh_float4 *d_verts, const vh_float4 *d_normals);
    /* any 16-byte {x,y,z,w} float4 (HIP's float4 included) is accepted as is */
    template <class F4>
    void integrate(const float4x4 &deltaT, const F4 *d_verts, const F4 *d_normals)
    {
        static_assert(sizeof(F4) == sizeof(vh_float4), "vertex map elements must be 16-byte float4");
        integrate(deltaT, reinterpret_cast<const vh_float4 *>(d_verts), reinterpret_cast<const vh_float4 *>(d_normals));
    }
    /* the same frame straight from the uint16 sensor image (preProcess + integrate in one call, no vertex
     * map in memory; Application.cpp:73-74,84); kInv: row-major 3x3 */
    void integrate(const float4x4 &deltaT, const uint16_t *d_depth, const float kInv[9]);
    void raycast(const float4x4 &pose, float *d_depth_out, float zNear = 0.1f, float zFar = 5.0f);
    /* raycast(pose, d_depth_out, d_normal_out) -- depth and, from the same pass, the camera-frame
     * normal of every hit (TSDF gradient; vh_raycast_normals) */
    void raycast(const float4x4 &pose, float *d_depth_out, vh_float4 *d_normal_out, float zNear = 0.1f, float zFar = 5.0f);
    /* depth plus camera-frame vertex and normal maps of the view (what CameraTracking::Align takes as target) */
    void raycast(const float4x4 &pose, float *d_depth_out, vh_float4 *d_vertices_out, vh_float4 *d_normals_out,
                 float zNear = 0.1f, float zFar = 5.0f);
    /* SDFRenderer::drawToFrontAndBack (SDFRenderer.cpp:165-208): nearest front / farthest back face of the
     * allocated blocks' cubes per pixel, as two depth images (vh_render_blocks) */
    void renderBlocks(const float4x4 &pose, float *d_front, float *d_back, float zNear = 0.1f, float zFar = 5.0f);
    void registerGLtoCUDA(SDFRenderer &) {}
    void unmapCUDApointers() {}

    int occupiedBlockCount();                          /* synchronises */
    /* README.md:15 lists deletion as a feature; the reference's deleteVoxelEntry
     * (VoxelUtils.cu:544-604) is never called.  Frees every block the last frame saw that holds
     * nothing within sdfThreshold of a surface (vh_garbage_collect). */
    void garbageCollect(float sdfThreshold);
    /* Opt-in extensions (voxelhash.h, vh_set_option): "pipeline" (one launch per frame, the commit and
     * TSDF update of a frame ride in the launch of the next; flush() launches the pending half),
     * "overflow_list", "band_mode", "depth_truncation", "weight_sample", "flatten_variant" (4: the walk-free frame -- the
     * occupancy-index walk in place of flattenKernel's scan of every entry: same results, 2-10x the frame rate on large tables), ... */
    void setOption(const char *name, int value);
    void setAllocBand(float bandMetres);
    void flush();
    /* count frames in count + 1 launches (vh_integrate_batch): poses = count * 16 row-major floats */
    void integrateBatch(int count, const float *poses, const vh_float4 *const *d_verts, const vh_float4 *const *d_normals);
    void setStream(void *hipStream);
    vh_context *context() { return ctx_; }
    const HashTableParams &params() const { return h_hashtableParams; }
};

#endif
