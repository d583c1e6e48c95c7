/*
 * voxelhash.h -- C-ABI of libvoxelhash_hip.so: the MI355X (gfx950) voxel-hashing
 * TSDF fusion path.  Plain C, plain pointers and sizes; no torch / C++ types.
 *
 * The library replaces the CUDA launcher layer of nilspin/VoxelHashing_demo
 * (VoxelUtils.h:5-13, implemented in VoxelUtils.cu) that SDF_Hashtable.cpp
 * binds to.  Two surfaces are exported:
 *
 *   1. vh_*  -- the explicit-context API (one table per context, one context
 *      per GPU, explicit stream, int status returns).  Everything else is
 *      built on it.
 *   2. the reference's own nine names (section "drop-in names" below) acting
 *      on a process-global default context, so that SDF_Hashtable.cpp links
 *      against this library unchanged apart from passing params by pointer.
 *
 * Device pointers are `hipMalloc`-class addresses (torch CUDA tensors'
 * data_ptr() qualify).  All device work is enqueued on the context's stream
 * and is asynchronous unless a function says it synchronises.  The usual
 * stream contract applies: a buffer handed to a call must be ready ON THAT
 * STREAM (produced there, or ordered before it with an event / a
 * synchronisation), and results are ready on that stream; the library adds
 * no synchronisation between streams.
 *
 * Paths in citations are relative to the reference checkout.
 */
#ifndef VOXELHASH_H
#define VOXELHASH_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ */
/* records -- layout identical to VoxelDataStructures.h                */
/* ------------------------------------------------------------------ */

/* VoxelDataStructures.h:12-17; 8 bytes */
typedef struct Voxel { float sdf; float weight; } Voxel;

/* VoxelDataStructures.h:20-26; 20 bytes, 4-byte aligned (the __align__(16)
 * placed before `struct` is ignored by the compilers) */
typedef struct VoxelEntry {
    int32_t pos[3];   /* int3 pos: block coordinate */
    int32_t ptr;      /* first voxel index of the block in the volume, -1 = free */
    int32_t offset;   /* overflow chain link, measured from the home bucket's last slot; 0 = none (always 0
                         unless the option "overflow_list" is on: the list code is dead in the reference) */
} VoxelEntry;

/* VoxelDataStructures.h:29-52; 176 bytes.  Matrices are row-major float4x4. */
typedef struct HashTableParams {
    float    global_transform[16];
    float    inv_global_transform[16];
    uint32_t numBuckets;
    uint32_t bucketSize;
    uint32_t attachedLinkedListSize;   /* iterations of the chain lookup loop (option "overflow_list"; dead code in the reference) */
    uint32_t numVoxelBlocks;
    int32_t  voxelBlockSize;           /* must be 8 */
    float    voxelSize;
    uint32_t numOccupiedBlocks;
    float    maxIntegrationDistance;   /* unused by the reference kernels */
    float    truncScale;               /* unused by the reference kernels (option "depth_truncation") */
    float    truncation;
    uint32_t integrationWeightSample;  /* unused by the reference kernels (option "weight_sample") */
    float    integrationWeightMax;
} HashTableParams;

/* float4 as CUDA/HIP lay it out */
typedef struct vh_float4 { float x, y, z, w; } vh_float4;

/* VoxelDataStructures.h:54-63 -- raw device pointers of one table.  The bucket
 * lock is an 8-byte epoch-stamped claim word per bucket instead of the
 * reference's memset-per-frame int (see DESIGN.md "bucket lock"): epoch in the
 * top 10 bits, then the winner's inverted launch rank, slot and record index. */
typedef struct PtrContainer {
    uint32_t   *d_heap;
    VoxelEntry *d_hashTable;
    VoxelEntry *d_compactifiedHashTable;
    uint64_t   *d_hashTableBucketMutex;
    Voxel      *d_SDFBlocks;
    int32_t    *d_heapCounter;
    int32_t    *d_compactifiedHashCounter;
} PtrContainer;

#define VH_FREE_BLOCK   (-1)          /* VoxelUtils.cu:19 */
#define VH_POS_SENTINEL 0x7fffffff    /* free-slot pos, VoxelUtils.cu:157 on a saturating cvt */

/* ------------------------------------------------------------------ */
/* status codes (the reference exits the process on any CUDA error,    */
/* helper_cuda.h:966-977; this ABI returns a code instead)             */
/* ------------------------------------------------------------------ */
enum {
    VH_OK = 0,
    VH_ERR_INVALID_ARGUMENT = 1,
    VH_ERR_NO_DEVICE = 2,        /* no usable HIP device */
    VH_ERR_OUT_OF_MEMORY = 3,
    VH_ERR_HIP = 4,              /* any other HIP runtime failure; see vh_last_error() */
    VH_ERR_NOT_INITIALISED = 5,  /* drop-in call before deviceAllocate() */
    VH_ERR_SINGULAR = 6,         /* vh_icp_solve: J^T J is not positive definite */
    VH_ERR_TIMEOUT = 7           /* vh_icp_align: a workgroup of the one-launch Align gave up waiting for the others (the call may be
                                  * repeated).  Frames: workgroups of a serialised one-launch frame (overflow list, option "pipeline_overflow") gave up waiting
                                  * for the pending frame's commit phase (option "spin_limit"): frames queued since the last successful
                                  * synchronisation have lost work.  Returned ONCE, by the first call that synchronises with the host and
                                  * sees the counter (vh_synchronize, vh_download*, vh_dist_flush); vh_get_counters reports the count
                                  * (spin_timeouts) without failing.  From then on the context runs such frames as two launches. */
};

/* projection / transform semantics */
enum {
    VH_SEM_REFERENCE = 0,  /* bit-faithful to the reference, quirks included: the projection
                              matrix is K transposed (common.h:16 through
                              cuda_SimpleMatrixUtil.h:316-320), blockInFrustum uses
                              global_transform (VoxelUtils.cu:348), the inverse pose is applied
                              to voxel indices and truncated (VoxelUtils.cu:797-800) */
    VH_SEM_PINHOLE = 1     /* physically meaningful variant used for benchmarks: K, inverse
                              pose in the frustum test plus z > 0, inverse pose in metres */
};

typedef struct vh_config {
    HashTableParams params;   /* as filled by SDF_Hashtable.cpp:62-73 */
    int32_t width;            /* depth image size; the reference hard-wires 640x480 */
    int32_t height;
    int32_t semantics;        /* VH_SEM_* */
    int32_t device;           /* HIP device ordinal, -1 = current device */
} vh_config;

typedef struct vh_counters {
    int32_t  occupied;          /* entries in the compact table (last flatten) */
    int32_t  heap_counter;      /* index of the top free heap slot; -1 = heap empty */
    uint32_t allocated_total;   /* blocks handed out since creation */
    uint32_t heap_exhausted;    /* insertions refused because the heap was empty */
    uint32_t candidates;        /* contenders of the last allocBlocks (demanded, even if the list was full) */
    uint32_t epoch;             /* bucket-lock epoch (= frames since creation) */
    uint32_t bin_overflow;      /* received key bins that exceeded their capacity (keys lost) */
    uint32_t freed_total;       /* blocks returned to the heap since creation (deletion / GC) */
    uint32_t last_freed;        /* ... by the last vh_delete_blocks / vh_garbage_collect */
    uint32_t cand_overflow;     /* contenders dropped since creation because the candidate list of their
                                   frame was full (their keys retry next frame; see "cand_capacity") */
    uint32_t spin_timeouts;     /* ERROR REPORT: workgroups of a serialised one-launch frame (option "overflow_list" with
                                   "pipeline" / vh_integrate_batch / vh_apply_frames_batch) that gave up waiting for the
                                   pending frame's commit phase after "spin_limit" polls (default 2^20, ~1.5 s): that frame
                                   is incomplete; once this is seen here the context runs its overflow-list frames as two
                                   launches each.  0 in every run so far. */
} vh_counters;

/* per-kernel device time, accumulated while profiling is on (HIP events on
 * the context's stream) */
typedef struct vh_kernel_times {
    uint64_t launches;          /* frames accumulated */
    double   alloc_claim_ms;
    double   alloc_commit_ms;
    double   flatten_ms;
    double   integrate_ms;
    double   raycast_ms;
    uint64_t raycast_launches;
    double   frame_scan_claim_ms;        /* fused vh_integrate, launch 1: claim || table walk */
    double   frame_commit_integrate_ms;  /* fused vh_integrate, launch 2: commit + TSDF update */
    double   view_export_ms;             /* vh_export_views: select walk + record packing */
    double   view_import_ms;             /* vh_import_view: clear + insert */
    double   gc_ms;                      /* vh_delete_blocks / vh_garbage_collect, all launches */
    uint64_t gc_calls;
    double   render_blocks_ms;           /* vh_render_blocks, all launches */
    double   frame_pipelined_ms;         /* pipelined frames: the one launch per frame */
} vh_kernel_times;

typedef struct vh_context vh_context;

/* what vh_download copies */
enum {
    VH_BUF_HASH_TABLE = 0,   /* numBuckets*bucketSize VoxelEntry */
    VH_BUF_COMPACT = 1,      /* numBuckets*bucketSize VoxelEntry (first `occupied` valid) */
    VH_BUF_SDF_BLOCKS = 2,   /* numVoxelBlocks*512 Voxel */
    VH_BUF_HEAP = 3          /* numVoxelBlocks uint32 */
};

/* ------------------------------------------------------------------ */
/* explicit-context API                                                */
/* ------------------------------------------------------------------ */

void vh_default_params(HashTableParams *p);          /* common.h:39-50 */
const char *vh_error_string(int code);
const char *vh_last_error(void);                     /* text of the last failure on this thread */
int  vh_device_count(void);                          /* 0 when no GPU is present */

/* SDF_Hashtable::SDF_Hashtable + deviceAllocate + calculateKinectProjectionMatrix
 * (SDF_Hashtable.cpp:60-81, VoxelUtils.cu:169-231).  Also owns the compact
 * table, its counter and the zero-initialised SDF volume, which the reference
 * borrows from OpenGL (SDFRenderer.cpp:34-61). */
int vh_create(const vh_config *cfg, vh_context **out);
int vh_destroy(vh_context *ctx);                     /* deviceFree, VoxelUtils.cu:213-222 */

int vh_set_stream(vh_context *ctx, void *hip_stream);   /* NULL = default stream */
int vh_set_projection(vh_context *ctx, const float m[9]);            /* row-major 3x3 */
int vh_set_raycast_intrinsics(vh_context *ctx, float fx, float fy, float cx, float cy);

/* Opt-in truncation-band allocation (the reference has it commented
 * out, VoxelUtils.cu:632-703): with band > 0 every valid pixel demands the blocks of
 * 2*ceil(band/step)+1 points on its viewing ray at camera depths z + (k-half)*step, step =
 * 4 voxels; the middle sample is the surface point itself.  0 (default) = the reference's
 * surface-block-only allocation.  One insertion per bucket per frame still holds. */
int vh_set_alloc_band(vh_context *ctx, float band_metres);

/* Opt-in extensions, all off by default (= the live reference path), chosen with vh_set_option:
 *   "overflow_list" 1   the bucket overflow list the reference carries as dead code (#ifdef LINKED_LIST_ENABLED,
 *                       VoxelUtils.cu:384-411 lookup, :458-539 insert, :578-602 delete): a key whose home bucket is
 *                       full goes to a free slot among the 9 slots behind the bucket (never another bucket's last
 *                       slot) and is chained from the home bucket's last slot through VoxelEntry::offset, at most
 *                       params.attachedLinkedListSize - 1 chained entries per bucket; both buckets are locked for
 *                       the frame; deletion leaves holes instead of compacting.  Must be set before the first
 *                       frame.  A shard's chains stay inside the shard.  With the list on, vh_alloc_blocks may run
 *                       once per lock epoch (several cameras per epoch: vh_insert_bins / vh_apply_frames_batch).
 *   "pipeline_overflow" one-launch frames ("pipeline", vh_integrate_batch, vh_apply_frames_batch) with the overflow list on are
 *                       serialised inside the launch, and every waiting workgroup pays a cache invalidate: 1 (default) =
 *                       taken only for small launches (up to 512 claim + walk workgroups), larger ones run as two launches
 *                       (C2: 21 us against 53 us); 0 = never; 2 = always.  Results are the same bits either way.
 *   "band_mode"         VH_BAND_RAY (default: vh_set_alloc_band's samples along the viewing ray) or
 *                       VH_BAND_NORMAL_DDA: every block the segment from p - band*n to p + band*n crosses, by a
 *                       block DDA (commented out in the reference, VoxelUtils.cu:632-703); n comes from the
 *                       d_normals argument of vh_alloc_blocks / vh_integrate (preProcess's normal map, camera
 *                       frame), pixels without a normal demand their surface block only.  Not offered by
 *                       vh_integrate_depth and the key-generation calls (they carry no normal map).
 *                       VH_BAND_RAY_DDA: the same block DDA along the pixel's VIEWING RAY -- every block the segment
 *                       from the ray's point at camera depth z - band to its point at z + band crosses (a band that
 *                       would begin behind the camera begins at the surface point): the exact set the samples of
 *                       VH_BAND_RAY approximate, at one transform per segment end and the DDA's divisions per pixel
 *                       instead of a transform and four divisions per sample; needs no normals, offered by every
 *                       entry point.
 *   "depth_truncation" 1   truncation + truncScale * depth in the TSDF update (VoxelUtils.cu:815, getTruncation)
 *   "weight_sample" 1      sample weight max(integrationWeightSample * 1.5 * (1 - (depth - 0.5) / 4.5), 1) instead of
 *                          0.1 (VoxelUtils.cu:808-811, :827) */
#define VH_BAND_RAY        0
#define VH_BAND_NORMAL_DDA 1
#define VH_BAND_RAY_DDA    2

/* SDF_Hashtable.cpp:15-21: stores the pose and its cofactor inverse
 * (cuda_SimpleMatrixUtil.h:944-1069, same summation order, fp32, on the host) */
int vh_set_pose(vh_context *ctx, const float pose[16]);

/* resetHashTableMutexes (VoxelUtils.cu:146-149): bumps the lock epoch instead of clearing 4*numBuckets bytes */
int vh_reset_mutexes(vh_context *ctx);
/* allocBlocks (VoxelUtils.cu:708-716, kernel :606-705): W*H float4 each, d_normals unused as in the reference (:631) */
int vh_alloc_blocks(vh_context *ctx, const vh_float4 *d_verts, const vh_float4 *d_normals);
/* flattenIntoBuffer (VoxelUtils.cu:751-768, kernel :719-749): with occupied_out != NULL it
 * synchronises the stream and returns the count like the reference does; with NULL it stays
 * asynchronous */
int vh_flatten(vh_context *ctx, int32_t *occupied_out);
/* integrateDepthMap (VoxelUtils.cu:844-852, kernel :790-842) over the compact list of the last flatten */
int vh_integrate_depth_map(vh_context *ctx, const vh_float4 *d_verts);

/* SDF_Hashtable::integrate (SDF_Hashtable.cpp:11-40) as one asynchronous call:
 * pose -> epoch bump -> allocBlocks -> flatten -> integrateDepthMap, no host
 * synchronisation and no device->host copy. */
int vh_integrate(vh_context *ctx, const float pose[16],
                 const vh_float4 *d_verts, const vh_float4 *d_normals);

/* The same frame straight from the uint16 sensor image (5000 units = 1 m, 0 = no measurement):
 * preProcess's vertex computation (CameraTrackingUtils.cu:63-73) runs inside the claim phase and the
 * TSDF update reads the image, so no vertex map exists in memory: 2 bytes per pixel in instead of
 * 16.  Equals vh_preprocess + vh_integrate bit for bit.  k_inv: row-major 3x3. */
int vh_integrate_depth(vh_context *ctx, const float pose[16], const uint16_t *d_depth, const float k_inv[9]);

/* Pipelined frames.  With vh_set_option(ctx, "pipeline", 1) a frame is ONE launch: vh_integrate enqueues
 * {claim || walk} of its frame together with the {commit + TSDF update} of the PREVIOUS frame, whose
 * results it leaves pending; the pending half is launched by the next vh_integrate / vh_integrate_depth,
 * by vh_flush, and by every call that reads or changes the model (counters, download, raycast, collection,
 * snapshot, step-level calls, vh_set_stream, vh_synchronize ...), so the library's own entry points always
 * see completed frames.  Code that reads the model through raw device pointers (vh_get_device_pointers)
 * calls vh_flush first.  The caller's depth / vertex buffer is only read by the launch of its own frame
 * (the deferred half works from a private copy of the camera-z plane), so buffers may be reused as with
 * the unpipelined calls.  Results equal the unpipelined frames bit for bit, with one documented
 * difference: a frame whose new blocks outnumber the free blocks of the heap allocates none of them
 * (they count as heap_exhausted and retry), where vh_integrate serves as many as there are blocks.
 * With "overflow_list" the frames are still one launch each, but serialised inside it: the claim and walk workgroups of
 * the new frame start when the commit phase of the pending one has finished (and that phase serves as many winners as the
 * heap has blocks, like the unpipelined frame).  bucketSize > 16 runs unpipelined.  vh_integrate_batch / vh_integrate_depth_batch: `count` frames (poses: count*16 host
 * floats; d_verts / d_normals / d_depth: host arrays of `count` device pointers, d_normals may be NULL)
 * in count + 1 launches -- the pipeline switched on for the call and flushed at its end. */
int vh_flush(vh_context *ctx);
int vh_integrate_batch(vh_context *ctx, int32_t count, const float *poses, const vh_float4 *const *d_verts,
                       const vh_float4 *const *d_normals);
int vh_integrate_depth_batch(vh_context *ctx, int32_t count, const float *poses, const uint16_t *const *d_depth,
                             const float k_inv[9]);

/* Stand-in for SDFRenderer::render (SDFRenderer.cpp:210-255): one ray per pixel
 * from `pose`, camera depth of the first +/- zero crossing into d_depth_out
 * (width*height floats, 0 = no hit).  Spec: DESIGN.md "raycast".
 * Two traversals, chosen with vh_set_option(ctx, "raycast_mode", ...):
 *   VH_RAYCAST_DDA (default)  the voxel DDA the reference's shader intends (raycastSDF.frag:121-177,
 *       Amanatides-Woo between the ray's two ends): every voxel the ray passes through between t_min and
 *       t_max, in order, each a sample with the voxel's own {sdf, weight} placed at the camera depth of the
 *       voxel's centre; crossing times are functions of the integer voxel coordinate (no accumulated tMax),
 *       ties as at :156-170; absent blocks and empty 4x4x4-block cells are left in one exact step.
 *   VH_RAYCAST_FIXED_STEP     rounds 1-2: samples at camera depth t_min + i*voxelSize, nearest voxel each.
 * Both: first pair of consecutive valid samples (block allocated, weight > 0) with sdf_prev > 0 >= sdf_cur,
 * linear interpolation.  The DDA refuses views of more than 2^22 voxel steps per ray.
 * vh_raycast_normals (DDA only) also writes, in the same pass, the normal of every hit: the TSDF gradient at
 * the second voxel of the pair (central differences where both neighbours are valid, one-sided otherwise),
 * normalised, in the CAMERA frame with w = 0 (the convention of calculateNormals, CameraTrackingUtils.cu:
 * 75-113); zeros for a miss or when an axis has no valid neighbour.
 * How the DDA is executed does not change a bit of the image; option "raycast_beam" picks the form: 2 = one block
 * list per 8x8-pixel wave, 1 = a walk per ray behind a per-wave beam front end, 0 = a walk per ray from t_min,
 * 3 (default) = by the view: 2 when 64 half-block slabs span [t_min, t_max] (coarse voxels), else 1. */
#define VH_RAYCAST_FIXED_STEP 0
#define VH_RAYCAST_DDA        1
int vh_raycast(vh_context *ctx, const float pose[16], float t_min, float t_max,
               float *d_depth_out);
int vh_raycast_normals(vh_context *ctx, const float pose[16], float t_min, float t_max,
                       float *d_depth_out, vh_float4 *d_normals_out);

/* Block silhouettes: the reference's one working render pass (SDFRenderer::drawToFrontAndBack,
 * SDFRenderer.cpp:165-208: one cube per entry -- block k covers world [8k, 8k+8]*voxelSize,
 * Application.cpp:130-132 -- nearest front face per pixel; notes.md:3-16 adds the back faces).  Per
 * pixel the camera depth at which its ray enters the nearest and leaves the farthest cube of ANY
 * allocated block, clipped to [t_min, t_max], by an exact ray/box test; 0 = no block.  Two W*H float
 * images.  Uses the raycast intrinsics. */
int vh_render_blocks(vh_context *ctx, const float pose[16], float t_min, float t_max, float *d_front,
                     float *d_back);

/* The compact table (d_compactifiedHashTable, VH_BUF_COMPACT) as the reference leaves it -- `occupied` dense
 * entries from index 0 -- is what vh_download, vh_get_device_pointers, vh_flush and vh_synchronize hand over:
 * inside a fused frame the list is kept with two ends (two counters instead of one hot word) and these calls
 * fold it first.  The addresses in a PtrContainer never change during the life of a context (fetch it once, as
 * the reference does, VoxelUtils.cu:141-148): pipelined frames alternate between two compact buffers and two
 * claim arrays internally, and vh_flush / vh_synchronize leave the dense list in the buffer
 * d_compactifiedHashTable names; d_hashTableBucketMutex names the first of the two claim arrays (consecutive lock
 * epochs of pipelined frames stake their claims in the two arrays alternately; every word carries its epoch). */
int vh_synchronize(vh_context *ctx);
int vh_get_counters(vh_context *ctx, vh_counters *out);              /* synchronises */
int vh_get_params(vh_context *ctx, HashTableParams *out);
int vh_get_device_pointers(vh_context *ctx, PtrContainer *out);
int vh_download(vh_context *ctx, int which, void *host_dst, size_t bytes);  /* synchronises */
/* the same for `bytes` bytes starting `offset_bytes` into the buffer (one 4 KiB block of a
 * multi-gigabyte volume: offset = 8 * entry.ptr) */
int vh_download_range(vh_context *ctx, int which, size_t offset_bytes, void *host_dst, size_t bytes);

/* DIAGNOSTICS AND TEST FACILITIES.  They are part of the library a deployment loads -- the tests and the profiles run on the
 * product, not on a build of their own -- and are supported as documented here; none of them changes a result:
 *   vh_debug_eval, vh_debug_set_raycast_stamps, vh_debug_occupy (below); the loop-back transport of voxelhash_dist.h
 *   (vh_dist_loopback_id: the N-rank exchange inside one process); vh_set_profiling / vh_get_kernel_times; the options
 *   "spin_limit"; environment: VOXELHASH_ROCTX=1 (roctx ranges named after the entry points -- vh_integrate, vh_integrate_depth,
 *   vh_flush, vh_raycast, vh_render_blocks, vh_icp_align, vh_garbage_collect, vh_preprocess, vh_dist_step_batch, vh_dist_raycast --
 *   for `rocprofv3 --marker-trace`; libroctx64 is loaded at run time, nothing is linked), VOXELHASH_LOOPBACK_TIMEOUT_S (how long a
 *   loop-back rank waits for its peers), VH_ICP_BLOCKS
 *   (workgroups of an ICP round), VOXELHASH_SEMANTICS (the drop-in names' semantics).
 * Code that exists only in diagnostics BUILDS (make EXTRA=-D...) and in no shipped library: VH_DEBUG_SKIP_ROLES (roles of the
 * pipelined launch return at once), VH_CLAIM_STAMPS, VH_DEBUG_DIST_* (per-phase time stamps and switch-offs). */
/* test hook: evaluates the device scalar helpers on n points; writes 8 int32 per
 * point: block x,y,z, hash, blockInFrustum, project() x,y, float->int of .w */
int vh_debug_eval(vh_context *ctx, const vh_float4 *d_points, int32_t n, int32_t *d_out);
/* diagnostics hook: the DDA raycast records per wave {start, end (100 MHz clock), voxel steps + jumps of lane 0,
 * pixel patch x | y << 16} in d_stamps (4 uint64 per wave, workgroups in launch order); NULL = off */
int vh_debug_set_raycast_stamps(vh_context *ctx, void *d_stamps);
/* test hook: `workgroups` x 256 lanes that stay resident for `microseconds` on `stream` (a device busy with another kernel) */
int vh_debug_occupy(vh_context *ctx, void *stream, int32_t workgroups, int32_t microseconds);

/* Options (18 names; anything else is rejected with VH_ERR_INVALID_ARGUMENT).  Results never depend on the tuning ones.
 *   semantics of the model (extensions of the reference, each with its oracle counterpart):
 *     "overflow_list" (0 | 1, before the first frame), "band_mode" (VH_BAND_*), "depth_truncation", "weight_sample" (0 | 1)
 *   the frame:
 *     "flatten_variant"  3 = the reference's walk over every VoxelEntry (flattenKernel, VoxelUtils.cu:719-749), 4 = the walk
 *                        over the bucket-occupancy bitmap and the non-empty buckets (same compact SET, the list's order is free
 *                        in the reference too: an atomic race, VoxelUtils.cu:737-746) -- the default: 2-11 x the frames/s of the
 *                        reference's walk at every table size measured
 *     "pipeline"         1: one launch per frame (a frame's commit + TSDF update ride in the next frame's launch)
 *     "pipeline_overflow" 0 | 1 | 2: one-launch frames with the overflow list never / by the launch's size / always
 *     "pipeline_shards"  0 | 1 | 2: the same for a shard's multi-camera frames (vh_apply_frames_batch)
 *     "fused_frame"      1 (default): vh_integrate as two launches; 0: the four step kernels of the step-level entry points
 *     "walk_nt"          non-temporal loads in the reference walk (default: on when the table exceeds the 256 MiB Infinity Cache)
 *     "integrate_grid", "commit_blocks"   workgroups of the TSDF update / of the commit phase in the two-launch frame
 *     "gen_frames_per_launch"  1..8 (default 4): frames of a batch one key-generation launch of vh_generate_keys*_batch takes
 *     "spin_limit"       polls a workgroup of a serialised one-launch frame waits for the pending commit phase (0: default)
 *   the raycast:
 *     "raycast_mode" (VH_RAYCAST_DDA | VH_RAYCAST_FIXED_STEP), "raycast_beam" (0 | 1 | 2 | 3 = by the view, default)
 *   formats / test hooks:
 *     "packet_format" (VH_PACKET_F32 / VH_PACKET_U16, below), "cand_capacity" (a smaller candidate list: vh_counters.cand_overflow)
 * Variants that were measured and lost (the persistent walk, 8 entries per lane in the frame's walk, the generic build where a
 * lean one exists, the raycast as three launches, 16x4 ray patches, LDS staging of the TSDF update and of the raycast's
 * blocks ...) are not in the library: DESIGN_LOG.md names the commit that last held each. */
int vh_set_option(vh_context *ctx, const char *name, int value);
int vh_set_profiling(vh_context *ctx, int enabled);
int vh_get_kernel_times(vh_context *ctx, vh_kernel_times *out, int reset);  /* synchronises */

/* ---- bucket-range sharding (multi-GPU; DESIGN.md "sharding") ---- */

/* A "multi-camera frame" generalises SDF_Hashtable::integrate to R cameras whose
 * frames enter one logical table together: one lock epoch, all cameras'
 * allocations first (camera order, then launch order, decides who wins a
 * bucket), then flatten + TSDF update camera by camera.  With R = 1 it is the
 * reference's integrate().  The table is cut into bucket ranges, one per GPU. */
#define VH_MAX_CAMERAS 32
#define VH_BIN_PER_BATCH (-1)      /* frame_stride of the batched shard calls: one key bin per shard for the whole batch */
#define VH_PACKET_HEADER_FLOATS 32   /* camera packet: pose[16], inverse[16], then W*H camera-z */

/* Restrict this context to the buckets [lo, hi) of a logical table of
 * params.numBuckets buckets; only those buckets' storage is allocated. */
int vh_create_shard(const vh_config *cfg, uint32_t bucket_lo, uint32_t bucket_hi,
                    vh_context **out);
/* Key generation half of allocBlocks for the pose set with vh_set_pose: per valid
 * pixel the block key of the surface point, frustum-tested, runs of equal keys
 * collapsed per wavefront.  Records are int4 {x,y,z,rank}, rank = camera_id<<27 |
 * launch rank<<6 | band sample, binned by owning shard (owner = hash / ceil(numBuckets/num_shards)):
 * bin s = d_bins[s*bin_stride*4 ...], record 0 = {count,0,0,0}, records 1..count the
 * keys (count > capacity-1 = overflow).  d_packet (nullable) receives the camera
 * packet: pose, inverse pose, camera-z plane (VH_PACKET_HEADER_FLOATS + W*H floats). */
int vh_generate_keys(vh_context *ctx, const vh_float4 *d_verts, uint32_t camera_id,
                     int32_t num_shards, int32_t *d_bins, int32_t capacity, int32_t bin_stride,
                     float *d_packet);
/* Insert the keys of num_bins received bins (same layout) into this shard under the
 * current lock epoch (call vh_reset_mutexes first).  bin_stride / packet_stride: distance
 * between consecutive bins (in records) / packets (in floats); 0 = dense.  Strides let
 * several frames per camera travel in one collective (bins[src][frame], packets[cam][frame]). */
int vh_insert_bins(vh_context *ctx, const int32_t *d_bins, int32_t num_bins, int32_t capacity,
                   int32_t bin_stride);
/* flatten + TSDF update of this shard for num_cams camera packets (contiguous,
 * camera order): one walk over the shard's entries for all cameras, then one
 * pass per visible block applying the cameras that see it in order. */
int vh_integrate_packets(vh_context *ctx, int32_t num_cams, const float *d_packets,
                         size_t packet_stride);
/* Batched forms (one host call, fewest launches) for `batch` frames per camera and exchange.
 * Layouts: bin of shard/source s, frame b at d_bins[(s*bin_stride + b*frame_stride)*4];
 * packet of camera c, frame b at d_packets[c*packet_stride + b*packet_frame_stride].
 * A stride of 0 means dense (frame_stride = capacity, bin_stride = batch*frame_stride,
 * packet_frame_stride = 32 + W*H, packet_stride = batch*packet_frame_stride).
 * vh_generate_keys_batch: poses = batch*16 host floats, d_verts = host array of `batch`
 * device pointers; the packets written are this camera's (d_packets[b*packet_frame_stride]).
 * frame_stride = VH_BIN_PER_BATCH: ONE bin per shard/source for the whole batch (at d_bins[s*bin_stride*4], capacity
 * records for all `batch` frames together, batch <= 32): a record carries its frame index in the rank's camera bits
 * (the camera is the source of the bin), and the launch of frame b claims the records of frame b.  Same tables as
 * per-frame bins; the point is the exchange: a fixed-size bin must hold the worst case of what it may receive, and
 * the worst case of a batch is much closer to its mean than the worst case of a single frame is (vh_dist_* sizes a
 * per-batch bin at 1.5 x batch x W*H/16 / shards records: 3.7 MB of bins per rank and exchange at 8 ranks, batch 8,
 * 640x480, against 19.7 MB with per-frame bins of W*H/16).
 * vh_apply_frames_batch: for b = 0..batch-1: new lock epoch, insert the num_bins bins of
 * frame b, walk + TSDF update for the num_cams packets of frame b; equals vh_reset_mutexes +
 * vh_insert_bins + vh_integrate_packets per frame.  Launches, option "pipeline_shards":
 *   1 (default) one launch per frame ({claim || walk} of frame b with {commit + TSDF update} of frame
 *               b-1) plus one for the batch's last frame: batch + 1;
 *   2           that last half stays pending and rides in the first launch of the NEXT call (batch
 *               launches per call); it is launched by vh_flush and by every call that reads or changes the
 *               model, like a pipelined single-camera frame.  The packets of the batch's LAST frame must
 *               stay valid and unchanged until then (vh_dist_* keeps three buffer sets for this);
 *   0           two launches per frame ({claim || walk}, {commit + integrate}).
 * Tables with bucketSize > 16 and view tables always take two launches per frame; with "overflow_list" the frames of
 * options 1 and 2 are serialised inside their launch (see the pipelined single-camera frame). */
int vh_generate_keys_batch(vh_context *ctx, int32_t batch, const float *poses,
                           const vh_float4 *const *d_verts, uint32_t camera_id, int32_t num_shards,
                           int32_t *d_bins, int32_t capacity, int32_t bin_stride, int32_t frame_stride,
                           float *d_packets, size_t packet_frame_stride);
int vh_apply_frames_batch(vh_context *ctx, int32_t batch, const int32_t *d_bins, int32_t num_bins,
                          int32_t capacity, int32_t bin_stride, int32_t frame_stride, int32_t num_cams,
                          const float *d_packets, size_t packet_stride, size_t packet_frame_stride);
/* Camera packet formats (what vh_integrate_packets / vh_apply_frames_batch read; chosen per context
 * with vh_set_option(ctx, "packet_format", ...)).  Strides stay in 4-byte units.
 *   VH_PACKET_F32: 32 floats {pose, inverse pose} + W*H float camera-z plane (written by
 *                  vh_generate_keys / vh_generate_keys_batch when d_packet(s) is given)
 *   VH_PACKET_U16: 36 floats {pose, inverse pose, K_inv row 2, depth unit 5000} + the W*H uint16
 *                  sensor image (W*H even): half the bytes on the wire; the owner recomputes the
 *                  camera z as preProcess does, (K_inv row 2 . (x,y,1)) * (d / 5000), so the result
 *                  equals the float path on vertex maps made by vh_preprocess from the same image. */
#define VH_PACKET_F32 0
#define VH_PACKET_U16 1
/* Sensor-depth packets of `batch` frames of this camera (poses: batch*16 host floats; d_depth: host
 * array of `batch` device pointers to W*H uint16; packet of frame b at d_packets[b*packet_frame_stride],
 * 0 = dense = 36 + W*H/2).  Keys for the same frames: vh_generate_keys_batch with d_packets = NULL on
 * the vertex maps vh_preprocess made from these images. */
int vh_write_packets_u16_batch(vh_context *ctx, int32_t batch, const float *poses,
                               const uint16_t *const *d_depth, const float k_inv[9], float *d_packets,
                               size_t packet_frame_stride);

/* Keys AND sensor-depth packets of `batch` frames of this camera from the uint16 images alone (one
 * launch per 8 frames: vertices are computed in place, the packet is the header plus the image).
 * Layouts and strides as in vh_generate_keys_batch; d_packets may be NULL (keys only). */
int vh_generate_keys_depth_batch(vh_context *ctx, int32_t batch, const float *poses,
                                 const uint16_t *const *d_depth, const float k_inv[9], uint32_t camera_id,
                                 int32_t num_shards, int32_t *d_bins, int32_t capacity, int32_t bin_stride,
                                 int32_t frame_stride, float *d_packets, size_t packet_frame_stride);

/* ------------------------------------------------------------------ */
/* raycast over shards ("replicate the compact table +   */
/* visible blocks"; DESIGN.md section 6 "raycast")                      */
/* ------------------------------------------------------------------ */
/* A ray samples blocks of every shard, so the rank that renders a view first gathers the
 * blocks the view can touch: each shard exports them as records, the records travel
 * (all-to-all with per-destination counts), the renderer imports them into a private view
 * table and calls vh_raycast on it.  The selection is a conservative superset of the blocks
 * the rays of the view sample, so the result equals vh_raycast on the unsharded table bit
 * for bit. */
typedef struct vh_view_record {
    int32_t  pos[3];
    int32_t  reserved;
    Voxel    voxels[512];
} vh_view_record;                /* 4112 bytes */

/* One walk over this table (shard) for n_views views (poses: n_views*16 host floats,
 * camera->world; the pyramid is that of the raycast intrinsics, t_min..t_max): the allocated
 * entries view v can touch are written to d_records, view 0's records first, then view 1's
 * ... without gaps; d_counts[v] (device) receives the number view v selected.  At most
 * `capacity` records are written per view (a count above capacity reports the loss).
 * d_records must hold n_views*capacity records, 16-byte aligned.  n_views <= VH_MAX_CAMERAS. */
int vh_export_views(vh_context *ctx, const float *poses, int32_t n_views, float t_min, float t_max,
                    vh_view_record *d_records, int32_t capacity, int32_t *d_counts);
/* `view`: an unsharded context of the same numBuckets / bucketSize that never integrated a
 * frame (numVoxelBlocks may be 1).  Its table is emptied and then holds exactly the `count`
 * records; the voxels stay in d_records, which must stay valid and unchanged until the next
 * import.  vh_raycast(view, ...) then renders them.  Records that find their bucket full
 * are dropped and counted in vh_counters.bin_overflow (never happens for records exported
 * from one logical table of the same geometry). */
int vh_import_view(vh_context *view, const vh_view_record *d_records, int32_t count);

/* The same round without any host synchronisation.  vh_export_views_fixed: the view poses are DEVICE
 * memory (n_views*16 floats, e.g. straight out of an all-gather; n_views <= 16) and view v's records go
 * to the fixed slot range [v*capacity, (v+1)*capacity) of d_records, so the exchange has equal sizes known
 * to the host; d_counts[v] = records selected (above capacity: the excess was not written).
 * The count also travels inside the payload: the spare header word (`reserved`) of the first record of a
 * view's slot range holds it.  vh_import_views: `num_sources` slot ranges of `capacity` records each,
 * source s holding min(count_s, capacity) records, count_s = d_counts[s] (device) or, with d_counts =
 * NULL, that header word; what the sources selected beyond the
 * capacity is added to vh_counters.bin_overflow of the view context.  Costs bandwidth instead of
 * latency: the exchange moves num_sources*capacity records whatever the counts are. */
int vh_export_views_fixed(vh_context *ctx, const float *d_poses, int32_t n_views, float t_min, float t_max,
                          vh_view_record *d_records, int32_t capacity, int32_t *d_counts);
int vh_import_views(vh_context *view, const vh_view_record *d_records, int32_t num_sources, int32_t capacity,
                    const int32_t *d_counts);

/* ------------------------------------------------------------------ */
/* block deletion / garbage collection         */
/* ------------------------------------------------------------------ */
/* The reference lists deletion as a feature (README.md:15) but deleteVoxelEntry
 * (VoxelUtils.cu:544-604) is never called and frees the block of the first FREE slot it
 * meets.  Built as the paper does it (Niessner et al. 2013, 4.4), asynchronous on the context's
 * stream, in its own lock epoch:
 *   vh_delete_blocks: for each key {x,y,z,_} (device, n records of 4 int32) present in this
 *     table (shard): the 512 voxels are zeroed, ptr/512 goes back on the heap
 *     (removeSingleBlockInHeap, :336-341), the entry is removed and the later entries of its
 *     bucket move down in order (a bucket's entries stay a prefix of its slots, which
 *     insertVoxelEntry :421-456 and every lookup rely on).  Absent keys are skipped.
 *   vh_garbage_collect: the same for every entry of the compact list (the blocks the last frame
 *     saw) whose voxels have max weight == 0, or min |sdf| over the voxels with weight > 0
 *     >= sdf_threshold.
 * Both leave the compact list empty (occupied = 0 until the next frame); vh_counters.last_freed
 * / freed_total report what was freed. */
int vh_delete_blocks(vh_context *ctx, const int32_t *d_keys, int32_t n);
int vh_garbage_collect(vh_context *ctx, float sdf_threshold);

/* ------------------------------------------------------------------ */
/* model dump / checkpoint                     */
/* ------------------------------------------------------------------ */
/* Text dump in the format of SDFRenderer::printSDFdata (SDFRenderer.cpp:71-110, written to
 * SDF_dump.txt by the reference): occupied count, then per compact entry pos / ptr / offset
 * and 512 sdf values at 4 decimals.  Like the original, entry i is followed by voxels
 * [512*i, 512*i+512) of the volume, not by the block its ptr names.  Synchronises. */
int vh_dump_sdf_text(vh_context *ctx, const char *path);
/* Binary snapshot of the model (hash table, heap, counters, the 4 KiB block of every
 * allocated entry) and its restore into a context created with the same configuration;
 * fusing can continue after vh_load_snapshot as if never interrupted.  Both synchronise. */
int vh_save_snapshot(vh_context *ctx, const char *path);
int vh_load_snapshot(vh_context *ctx, const char *path);

/* ------------------------------------------------------------------ */
/* depth pre-processing                        */
/* ------------------------------------------------------------------ */
/* preProcess (CameraTrackingUtils.cu:115-120) = calculateVertexPositions (:50-73) +
 * calculateNormals (:75-113) as one kernel: uint16 depth (5000 units = 1 m, 0 = invalid)
 * -> float4 vertex map (K_inv*(x,y,1)*depth, w = 1; invalid -> (0,0,0,1)) and float4
 * normal map (central-difference cross product, normalised; 0 on the border or next to an
 * invalid pixel).  k_inv: the 3x3 the reference uploads with SetCameraIntrinsic, read
 * row-major.  The outputs are what vh_integrate takes as d_verts / d_normals. */
int vh_preprocess(const uint16_t *d_depth, const float k_inv[9], int32_t width, int32_t height,
                  vh_float4 *d_positions, vh_float4 *d_normals, void *hip_stream);
/* the reference's own names (CameraTrackingUtils.cu:218-222, 115-120): 640x480, default
 * stream, synchronous */
bool SetCameraIntrinsic(const float *intrinsic, const float *invIntrinsic);
void preProcess(vh_float4 *positions, vh_float4 *normals, const uint16_t *depth);

/* ------------------------------------------------------------------ */
/* camera tracking: frame-to-frame point-to-plane ICP                  */
/* (second half)                               */
/* ------------------------------------------------------------------ */
/* The reference's tracker (CameraTracking::Align, CameraTracking.cpp:27-69; never called,
 * Application.cpp:75) aligns the vertex map of one frame (input) to the vertex + normal maps of
 * another (target): per round FindCorrespondences (CameraTrackingUtils.cu:131-185) projects every
 * input point, moved by the current estimate, into the target image and keeps the pair when the
 * point-to-plane distance d = dot(q - t, n) is below the threshold; CalculateJacAndResKernel
 * (Solver.cu:40-54) writes the 6 x N Jacobian [n, t x n]; cublasSgemv / cublasSsyrk reduce it to
 * J^T r and J^T J (Solver.cpp:81-90); the host solves and composes in SE3 (Solver.cpp:104-106,
 * SE3.cpp).  Here one fused pass per round produces the 27 sums directly.  With the target maps
 * taken from vh_raycast + vh_depth_to_maps this is the frame-to-model tracking of KinectFusion. */
#define VH_ICP_ABS_DISTANCE 1   /* keep |d| < threshold (the reference keeps the signed d < threshold, :170) */
#define VH_ICP_NEED_TARGET  2   /* skip pixels whose target has no depth or no normal (the reference pairs them) */

typedef struct vh_icp vh_icp;    /* workspace for one image size on one device */
typedef struct vh_icp_system {
    double   JTJ[36];            /* row-major, symmetric; parameter order (v, w) = translation, rotation */
    double   JTr[6];
    double   error;              /* sum of d over the kept pairs (computeCorrespondences' return value) */
    uint32_t count;              /* kept pairs */
} vh_icp_system;

int vh_icp_create(int32_t width, int32_t height, int32_t device /* -1: current */, vh_icp **out);
int vh_icp_destroy(vh_icp *icp);
int vh_icp_set_stream(vh_icp *icp, void *hip_stream);
/* One round's linear system for the estimate `delta` (row-major 4x4 mapping input points into the
 * target's camera frame); K = row-major intrinsics.  Device maps are width*height float4.  fp32
 * sums in a fixed order (reproducible); synchronises to return the system. */
int vh_icp_build_system(vh_icp *icp, const vh_float4 *d_input, const vh_float4 *d_target,
                        const vh_float4 *d_target_normals, const float delta[16], const float K[9],
                        float dist_thres, int32_t flags, vh_icp_system *out);
/* The same, and also the three maps computeCorrespondences fills (zero where no pair is kept). */
int vh_icp_correspondences(vh_icp *icp, const vh_float4 *d_input, const vh_float4 *d_target,
                           const vh_float4 *d_target_normals, const float delta[16], const float K[9],
                           float dist_thres, int32_t flags, vh_float4 *d_corres, vh_float4 *d_corres_normals,
                           float *d_residuals, vh_icp_system *out);
/* Host only.  update = -(JTJ^-1 JTr) by an LDL^T factorisation, estimate = log(exp(update) exp(estimate)) with
 * twists (v, w) as in SE3.cpp:4-22.  VH_ERR_SINGULAR leaves the estimate untouched. */
int  vh_icp_solve(const vh_icp_system *sys, double estimate[6]);
void vh_se3_exp(const double twist[6], double T[16]);
void vh_se3_log(const double T[16], double twist[6]);
/* Up to max_iters rounds (the reference: 20) from the start value in `delta`, which receives the
 * result; stops early when the summed residual is exactly 0 (:52) or the system is singular.
 * All rounds run in ONE launch whose workgroups wait for each other between rounds
 * (VH_ICP_PERSISTENT=0 in the environment at vh_icp_create: one launch per round -- same result,
 * bit for bit).  The waits are bounded: VH_ERR_TIMEOUT when a
 * workgroup gave up (another kernel holding the chip for about a second), `delta` is then untouched. */
int vh_icp_align(vh_icp *icp, const vh_float4 *d_input, const vh_float4 *d_target,
                 const vh_float4 *d_target_normals, const float K[9], float dist_thres, int32_t max_iters,
                 int32_t flags, float delta[16], vh_icp_system *last, int32_t *iterations);
/* float depth image in metres (0 = nothing, e.g. vh_raycast's output) -> vertex + normal maps with
 * the arithmetic of preProcess; asynchronous on hip_stream. */
int vh_depth_to_maps(const float *d_depth, const float k_inv[9], int32_t width, int32_t height,
                     vh_float4 *d_positions, vh_float4 *d_normals, void *hip_stream);
/* vh_raycast, then vertex and normal maps (camera frame) of the same view, with the K^-1 of the
 * raycast intrinsics: the model as an ICP target (raycast(pose, depth, normals)). */
int vh_raycast_maps(vh_context *ctx, const float pose[16], float t_min, float t_max, float *d_depth_out,
                    vh_float4 *d_vertices_out, vh_float4 *d_normals_out);
/* One frame of the closed loop (the demo's frame order, Application.cpp:73-90, with the tracker switched on):
 * vh_preprocess(d_depth) -> vh_icp_align(input maps, model maps; start = identity) -> pose <- pose . delta (row-major
 * double 4x4, in / out) -> vh_integrate_depth(pose, d_depth) -> vh_raycast_maps(pose) into the model maps for the next
 * frame.  Everything runs on the context's stream (the tracker must be bound to it: vh_icp_set_stream); the one host
 * synchronisation is the one inside vh_icp_align.  The first frame of a sequence is vh_integrate_depth + vh_raycast_maps
 * at the start pose.  d_input_*: scratch maps the call fills. */
int vh_fusion_step(vh_context *ctx, vh_icp *icp, const uint16_t *d_depth, const float k_inv[9], const float K[9],
                   float dist_thres, int32_t max_iters, int32_t flags, float t_min, float t_max,
                   vh_float4 *d_input_vertices, vh_float4 *d_input_normals, float *d_model_depth,
                   vh_float4 *d_model_vertices, vh_float4 *d_model_normals, double pose[16],
                   vh_icp_system *last, int32_t *iterations);
/* the reference's own name (CameraTrackingUtils.cu:187-215; float4x4 by value there, a pointer to
 * its 16 row-major floats here): intrinsics from SetCameraIntrinsic, threshold 0.08 (common.h:12),
 * synchronous, returns the summed residual */
float computeCorrespondences(const vh_float4 *d_input, const vh_float4 *d_target,
                             const vh_float4 *d_targetNormals, vh_float4 *corres, vh_float4 *corresNormals,
                             float *residual, const float *deltaTransform, int width, int height);

/* ------------------------------------------------------------------ */
/* drop-in names (VoxelUtils.h:5-13); process-global default context    */
/* ------------------------------------------------------------------ */
/* The reference declares these with `const HashTableParams&`; a C++ reference
 * is a pointer at the ABI level, so the C declarations take a pointer.
 * Errors follow the reference convention: message on stderr + exit(1).
 * mapGLobjectsToCUDApointers (VoxelUtils.h:13) is not carried over: the
 * compact table / counter / volume are library-owned device buffers. */
void updateConstantHashTableParams(const HashTableParams *params);
void deviceAllocate(const HashTableParams *params);
void deviceFree(void);
void resetHashTableMutexes(const HashTableParams *params);
void allocBlocks(const vh_float4 *verts, const vh_float4 *normals);
int  flattenIntoBuffer(const HashTableParams *params);
void calculateKinectProjectionMatrix(void);
void integrateDepthMap(const HashTableParams *params, const vh_float4 *verts);
/* the default context behind the drop-in names (NULL before deviceAllocate) */
vh_context *vh_default_context(void);

#ifdef __cplusplus
}
#endif
#endif /* VOXELHASH_H */
