/*
 * vh_oracle.h -- CPU oracle for the voxel-hashing TSDF fusion hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a scalar C restatement of the reference
 * algorithm (nilspin/VoxelHashing_demo, VoxelUtils.cu + SDF_Hashtable.cpp).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call it; the product (libvoxelhash_hip.so) never does.
 *
 * PARITY STATUS: "parity unpinned" in the formal sense -- the reference ships
 * no tests, golden vectors or fixtures for this path and
 * cannot be built in this image (needs nvcc, the CUDA runtime, OpenGL 4.3,
 * SDL2, GLM; building it would require stand-ins for CUDA headers, which is
 * not allowed).  The oracle is anchored instead on the values SURVEY.md /
 * BASELINE.md recorded from a host emulation of the unmodified reference
 * source during the survey ([probe] values: hash KATs, rounding KATs, and the
 * block / voxel counts of two sphere scenes); tests/test_oracle_anchors.py
 * checks every one of them.
 *
 * Each function cites the reference file:line it follows (paths relative to
 * /root/reference).
 */
#ifndef VH_ORACLE_H
#define VH_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VHO_FREE_BLOCK   (-1)          /* VoxelUtils.cu:19 */
#define VHO_LOCKED_BLOCK (-2)          /* VoxelUtils.cu:20 */
#define VHO_POS_SENTINEL 0x7fffffff    /* (int)+inf on a saturating GPU cvt, VoxelUtils.cu:157 */

/* projection / transform semantics */
#define VHO_SEM_REFERENCE 0   /* bit-faithful quirks: K^T, global_transform in the frustum
                                 test, inverse pose applied in voxel-index units */
#define VHO_SEM_PINHOLE   1   /* physically meaningful: K, inverse pose in the frustum test
                                 (+ z > 0), inverse pose applied in metres */

/* VoxelDataStructures.h:12-17 -- 8 bytes */
typedef struct { float sdf; float weight; } vho_voxel;

/* VoxelDataStructures.h:20-26 -- 20 bytes, align 4 */
typedef struct { int32_t pos[3]; int32_t ptr; int32_t offset; } vho_entry;

/* VoxelDataStructures.h:29-52 -- 176 bytes; matrices row-major */
typedef struct {
    float    global_transform[16];
    float    inv_global_transform[16];
    uint32_t numBuckets;
    uint32_t bucketSize;
    uint32_t attachedLinkedListSize;
    uint32_t numVoxelBlocks;
    int32_t  voxelBlockSize;
    float    voxelSize;
    uint32_t numOccupiedBlocks;
    float    maxIntegrationDistance;
    float    truncScale;
    float    truncation;
    uint32_t integrationWeightSample;
    float    integrationWeightMax;
} vho_params;

typedef struct {
    uint32_t pixels_valid;       /* verts with z != 0 */
    uint32_t pixels_in_frustum;  /* ... whose block passes blockInFrustum */
    uint32_t inserted;           /* entries inserted this frame */
    uint32_t lock_losses;        /* contenders that met a locked bucket */
    uint32_t bucket_full;        /* pixels whose bucket had no free slot and no match */
    uint32_t heap_exhausted;     /* insertions refused because the heap was empty */
    uint32_t occupied;           /* compact count (allocated AND in frustum) */
    uint32_t voxels_updated;     /* voxels written by the TSDF update */
    int32_t  heap_counter;       /* heap counter after the frame */
} vho_frame_stats;

typedef struct vho_table vho_table;

/* ---- lifecycle (SDF_Hashtable.cpp:60-81, VoxelUtils.cu:169-231) ---- */
void       vho_default_params(vho_params *p);   /* common.h:39-50 values */
vho_table *vho_create(const vho_params *p, int width, int height, int semantics);
void       vho_destroy(vho_table *t);
void       vho_set_projection(vho_table *t, const float m[9]);  /* row-major 3x3 */
void       vho_set_raycast_intrinsics(vho_table *t, float fx, float fy, float cx, float cy);
/* opt-in truncation-band allocation; 0 = surface block only */
void       vho_set_alloc_band(vho_table *t, float band_metres);
#define VHO_BAND_RAY        0   /* samples on the viewing ray, half-block steps (round 1) */
#define VHO_BAND_NORMAL_DDA 1   /* block DDA from p - b*n to p + b*n (VoxelUtils.cu:632-703, commented out there) */
#define VHO_BAND_RAY_DDA    2   /* the same block DDA along the viewing ray, from depth z - b to z + b (round 4) */
void       vho_set_band_mode(vho_table *t, int mode);
void       vho_set_normals(vho_table *t, const float *normals);   /* W*H float4, camera frame; borrowed; NULL = none */
/* opt-in overflow linked list (VoxelUtils.cu:384-411, 458-539, 578-602: dead code there); chains wrap
 * inside segments of `segment_buckets` buckets (0 = the table's own bucket range) */
void       vho_set_overflow(vho_table *t, int enabled, uint32_t segment_buckets);
/* opt-in TSDF update variants the reference has commented out */
#define VHO_INT_DEPTH_TRUNCATION 1   /* truncation + truncScale * depth (VoxelUtils.cu:815, getTruncation :261-264) */
#define VHO_INT_WEIGHT_SAMPLE    2   /* weight = max(integrationWeightSample * 1.5 * (1 - depth01), 1) (:808-811, :827) */
void       vho_set_integrate_flags(vho_table *t, int flags);

/* ---- per-frame steps (SDF_Hashtable.cpp:11-40) ---- */
void vho_set_pose(vho_table *t, const float pose[16]);          /* + cofactor inverse */
void vho_reset_mutexes(vho_table *t);                            /* VoxelUtils.cu:146-149 */
void vho_alloc_blocks(vho_table *t, const float *verts);         /* VoxelUtils.cu:606-716 */
int  vho_flatten(vho_table *t);                                  /* VoxelUtils.cu:719-768 */
void vho_integrate_depth_map(vho_table *t, const float *verts);  /* VoxelUtils.cu:790-852 */
/* all of the above in the reference's order; returns the occupied count */
int  vho_integrate(vho_table *t, const float pose[16], const float *verts,
                   vho_frame_stats *stats);

/* the same frame on `threads` host threads (OpenMP), identical results; for bench.py's cpu_baseline */
int  vho_integrate_mt(vho_table *t, const float pose[16], const float *verts, int threads,
                      vho_frame_stats *stats);

/* ---- raycast (build spec; self-pinned) ---- */
/* fixed-step march (rounds 1-2): samples at camera depth t_min + i*voxelSize, nearest voxel each */
void vho_raycast(vho_table *t, const float pose[16], float t_min, float t_max,
                 float *depth_out /* W*H */);
/* voxel DDA (raycastSDF.frag:121-177): every voxel the ray passes through, in order; optional camera-frame
 * normals of the hits (W*H float4, w = 0; NULL = none).  jumps != 0 leaves absent blocks in one go -- the
 * image has the same bits either way (the accelerated form the HIP kernel mirrors). */
void vho_raycast_dda(vho_table *t, const float pose[16], float t_min, float t_max, int jumps,
                     float *depth_out /* W*H */, float *normal_out /* W*H*4 or NULL */);

/* ---- block silhouettes (SDFRenderer::drawToFrontAndBack) ---- */
void vho_render_blocks(const vho_table *t, const float pose[16], float t_min, float t_max, float *front, float *back);

/* ---- block deletion / garbage collection (build extension) ---- */
int vho_delete_blocks(vho_table *t, const int32_t *keys /* n x {x,y,z,_} */, int n);
int vho_garbage_collect(vho_table *t, float sdf_threshold);

/* ---- raycast over shards (build extension): the blocks a view can
 * touch are gathered from the shards into a view table that raycasts like the whole ---- */
#define VHO_VIEW_RECORD_BYTES 4112     /* {int32 pos[3], 0, 512 x {sdf, weight}} */
void vho_view_frustum(const vho_table *t, const float pose[16], float t_min, float t_max, float f[22]);
int  vho_view_holds_block(const vho_table *t, const float f[22], const int32_t key[3]);
int  vho_export_view(const vho_table *t, const float pose[16], float t_min, float t_max,
                     uint8_t *records, int capacity);
int  vho_import_view(vho_table *view, const uint8_t *records, int count);

/* ---- bucket-range sharding (build extension for multi-GPU) ---- */
vho_table *vho_create_shard(const vho_params *p, int width, int height, int semantics,
                            uint32_t bucket_lo, uint32_t bucket_hi);
int  vho_generate_keys(vho_table *t, const float *verts, uint32_t camera_id, int num_shards,
                       int32_t *bins, int capacity);
int  vho_insert_bins(vho_table *t, const int32_t *bins, int num_bins, int capacity);
void vho_write_packet(vho_table *t, const float *verts, float *packet);
int  vho_integrate_packets(vho_table *t, int num_cams, const float *packets);
uint32_t vho_bucket_lo(const vho_table *t);
uint32_t vho_bucket_hi(const vho_table *t);

/* ---- depth pre-processing (CameraTrackingUtils.cu:50-120) ---- */
void vho_preprocess(const uint16_t *depth, const float k_inv[9], int width, int height,
                    float *positions, float *normals);

/* ---- accessors ---- */
const vho_params *vho_get_params(const vho_table *t);
vho_entry        *vho_hash_table(vho_table *t);      /* numBuckets*bucketSize entries */
vho_entry        *vho_compact_table(vho_table *t);   /* first vho_compact_count valid */
int               vho_compact_count(const vho_table *t);
const uint32_t   *vho_heap(const vho_table *t);      /* numVoxelBlocks ids, [0, heap_counter] free */
vho_voxel        *vho_sdf_blocks(vho_table *t);      /* numVoxelBlocks*512 voxels */
int               vho_heap_counter(const vho_table *t);
const vho_frame_stats *vho_last_stats(const vho_table *t);

/* ---- frame-to-frame point-to-plane ICP (second half; vh_icp_oracle.c) ---- */
#define VHO_ICP_ABS_DISTANCE 1   /* |d| < threshold instead of the reference's signed d < threshold */
#define VHO_ICP_NEED_TARGET  2   /* skip pixels whose target has no depth / no normal */
void vho_depth_to_maps(const float *depth, const float k_inv[9], int W, int H, float *positions, float *normals);
void vho_icp_build_system(const float *input, const float *target, const float *target_normals,
                          const float delta[16], const float K[9], float dist_thres, int W, int H, int flags,
                          double JTJ[36], double JTr[6], double *error, uint32_t *count);
double vho_icp_correspondences(const float *input, const float *target, const float *target_normals,
                               const float delta[16], const float K[9], float dist_thres, int W, int H, int flags,
                               float *corres, float *corres_normals, float *residuals, uint32_t *count);
void vho_se3_exp(const double twist[6], double T[16]);
void vho_se3_log(const double T[16], double twist[6]);
int  vho_icp_solve(const double JTJ[36], const double JTr[6], double estimate[6]);
int  vho_icp_align(const float *input, const float *target, const float *target_normals, const float K[9],
                   float dist_thres, int W, int H, int max_iters, int flags, float delta[16], double *final_error,
                   uint32_t *final_count);

/* ---- scalar helpers, exported for known-answer tests ---- */
int32_t  vho_float2int_rz(float x);                               /* GPU cvt semantics */
uint32_t vho_hash(int32_t x, int32_t y, int32_t z, uint32_t numBuckets);
void     vho_world2voxel(const float p[3], float voxelSize, int32_t out[3]);
void     vho_voxel2block(const int32_t v[3], int32_t blockSize, int32_t out[3]);
void     vho_world2block(const float p[3], float voxelSize, int32_t blockSize, int32_t out[3]);
void     vho_invert4x4(const float m[16], float out[16]);
void     vho_mat4_mul_vec4(const float m[16], const float v[4], float out[4]);
void     vho_project(const float m[9], const float p[3], int32_t out[2]);
int      vho_block_in_frustum(const vho_table *t, const int32_t block[3]);
void     vho_combine_voxel(const vho_voxel *o, const vho_voxel *c, float wmax, vho_voxel *out);
uint32_t vho_launch_rank(int x, int y, int width);

#ifdef __cplusplus
}
#endif
#endif
