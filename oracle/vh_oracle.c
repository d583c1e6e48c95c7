/*
 * vh_oracle.c -- scalar CPU restatement of the reference voxel-hashing path.
 * TEST INFRASTRUCTURE ONLY; see vh_oracle.h for the parity status.
 *
 * Build: gcc -std=c99 -O2 -ffp-contract=off -fno-fast-math  (the reference is
 * built with nvcc -fmad=false, CMakeLists.txt:23, so no FMA contraction is
 * allowed anywhere in here).  x86-64 SSE arithmetic is IEEE binary32, the same
 * as the GPU's fp32 + - * / with denormals on.
 *
 * Determinism rule: the reference's bucket race is resolved
 * the way a sequential run of its own launch grid resolves it -- pixels are
 * visited 16x16-tile-major (allocBlocks, VoxelUtils.cu:710-714), the first
 * contender for a bucket in that order wins it for the frame.
 */
#include "vh_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define VHO_INF_F (__builtin_inff())

struct vho_table {
    vho_params p;
    int   width, height, semantics;
    float proj[9];                 /* "kinectProjectionMatrix", VoxelUtils.cu:24 */
    float rc_fx, rc_fy, rc_cx, rc_cy;
    uint32_t bucket_lo, bucket_hi;  /* this table owns buckets [lo, hi) of the logical table */
    float alloc_band;               /* 0: surface block only (reference); > 0: truncation-band allocation */
    int   band_mode;                /* VHO_BAND_RAY / VHO_BAND_NORMAL_DDA / VHO_BAND_RAY_DDA */
    const float *normals;           /* normal map of the running allocBlocks (NORMAL_DDA), camera frame */
    int   overflow;                 /* overflow linked list on (VoxelUtils.cu:384-411,458-539,578-602 done right) */
    uint32_t overflow_seg;          /* chains wrap inside segments of this many buckets (0: the owned range) */
    uint32_t epoch;                 /* lock epochs since creation (vho_reset_mutexes) */
    uint32_t *slot_epoch;           /* epoch in which a slot received its entry (overflow look-ahead rule) */
    int   integrate_flags;          /* VHO_INT_DEPTH_TRUNCATION | VHO_INT_WEIGHT_SAMPLE */

    uint32_t  *heap;               /* PtrContainer, VoxelDataStructures.h:54-63 */
    vho_entry *table;
    vho_entry *compact;
    int32_t   *mutex;
    vho_voxel *blocks;
    const vho_voxel *view_blocks;  /* vho_import_view: voxels live in the caller's records */
    int32_t    heap_counter;
    int32_t    compact_counter;

    vho_frame_stats stats;
};

/* ------------------------------------------------------------------ */
/* scalar helpers                                                      */
/* ------------------------------------------------------------------ */

/* float -> int as the GPU does it (cvt.rzi.s32.f32 / v_cvt_i32_f32):
 * truncate toward zero, saturate, NaN -> 0.  Used wherever the reference
 * writes int(f), make_int2(float,float) or __float2int_rz
 * (helper_math.h:160-163, VoxelUtils.cu:353-354,776,799). */
int32_t vho_float2int_rz(float x)
{
    if (x != x) return 0;
    if (x >= 2147483648.0f) return INT32_MAX;
    if (x <= -2147483648.0f) return INT32_MIN;
    return (int32_t)x;
}

static int32_t wrap_mul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
static int32_t wrap_add(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static int32_t wrap_sub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }

/* calculateHash, VoxelUtils.cu:250-259.  The modulo is evaluated in unsigned
 * arithmetic because numBuckets is unsigned (VoxelDataStructures.h:36); the
 * res<0 fix-up is dead. */
uint32_t vho_hash(int32_t x, int32_t y, int32_t z, uint32_t numBuckets)
{
    const int32_t p0 = 73856093, p1 = 19349669, p2 = 83492791;
    uint32_t h = (uint32_t)(wrap_mul(x, p0) ^ wrap_mul(y, p1) ^ wrap_mul(z, p2));
    return h % numBuckets;
}

/* world2Voxel, VoxelUtils.cu:280-287: true divide, then round half away from
 * zero through copysignf(1,p)*0.5 and a truncating conversion. */
void vho_world2voxel(const float p[3], float voxelSize, int32_t out[3])
{
    for (int k = 0; k < 3; ++k) {
        float q = p[k] / voxelSize;
        int32_t sgn = vho_float2int_rz(copysignf(1.0f, q));   /* make_int3(float..) */
        float half = (float)((double)sgn * 0.5);              /* int * 0.5 -> double -> float */
        out[k] = vho_float2int_rz(q + half);
    }
}

/* voxel2Block, VoxelUtils.cu:266-278: floor division for negative voxels. */
void vho_voxel2block(const int32_t v[3], int32_t size, int32_t out[3])
{
    for (int k = 0; k < 3; ++k) {
        int32_t c = v[k];
        if (c < 0) c = wrap_sub(c, size - 1);
        out[k] = c / size;
    }
}

/* world2Block, VoxelUtils.cu:306-309 */
void vho_world2block(const float p[3], float voxelSize, int32_t blockSize, int32_t out[3])
{
    int32_t v[3];
    vho_world2voxel(p, voxelSize, v);
    vho_voxel2block(v, blockSize, out);
}

/* float4x4::operator*(float4), cuda_SimpleMatrixUtil.h:888-896: each row is
 * summed left to right. */
void vho_mat4_mul_vec4(const float m[16], const float v[4], float out[4])
{
    float r[4];
    for (int i = 0; i < 4; ++i)
        r[i] = m[4*i+0]*v[0] + m[4*i+1]*v[1] + m[4*i+2]*v[2] + m[4*i+3]*v[3];
    memcpy(out, r, sizeof r);
}

/* float3x3::operator*(float3), cuda_SimpleMatrixUtil.h:482-488 */
static void mat3_mul_vec3(const float m[9], const float v[3], float out[3])
{
    float r[3];
    for (int i = 0; i < 3; ++i)
        r[i] = m[3*i+0]*v[0] + m[3*i+1]*v[1] + m[3*i+2]*v[2];
    memcpy(out, r, sizeof r);
}

/* float4x4::getInverse, cuda_SimpleMatrixUtil.h:944-1069: adjugate by cofactor
 * expansion.  The table lists, for every output element, the six signed
 * triple products in the order the reference sums them, so the fp32 result
 * is the same bit pattern.  -a*b*c == -(a*b*c) and x - y == x + (-y) exactly. */
static const signed char k_cof[16][6][4] = {
    /* out  0 */ {{+1,5,10,15},{-1,5,11,14},{-1,9,6,15},{+1,9,7,14},{+1,13,6,11},{-1,13,7,10}},
    /* out  1 */ {{-1,1,10,15},{+1,1,11,14},{+1,9,2,15},{-1,9,3,14},{-1,13,2,11},{+1,13,3,10}},
    /* out  2 */ {{+1,1,6,15},{-1,1,7,14},{-1,5,2,15},{+1,5,3,14},{+1,13,2,7},{-1,13,3,6}},
    /* out  3 */ {{-1,1,6,11},{+1,1,7,10},{+1,5,2,11},{-1,5,3,10},{-1,9,2,7},{+1,9,3,6}},
    /* out  4 */ {{-1,4,10,15},{+1,4,11,14},{+1,8,6,15},{-1,8,7,14},{-1,12,6,11},{+1,12,7,10}},
    /* out  5 */ {{+1,0,10,15},{-1,0,11,14},{-1,8,2,15},{+1,8,3,14},{+1,12,2,11},{-1,12,3,10}},
    /* out  6 */ {{-1,0,6,15},{+1,0,7,14},{+1,4,2,15},{-1,4,3,14},{-1,12,2,7},{+1,12,3,6}},
    /* out  7 */ {{+1,0,6,11},{-1,0,7,10},{-1,4,2,11},{+1,4,3,10},{+1,8,2,7},{-1,8,3,6}},
    /* out  8 */ {{+1,4,9,15},{-1,4,11,13},{-1,8,5,15},{+1,8,7,13},{+1,12,5,11},{-1,12,7,9}},
    /* out  9 */ {{-1,0,9,15},{+1,0,11,13},{+1,8,1,15},{-1,8,3,13},{-1,12,1,11},{+1,12,3,9}},
    /* out 10 */ {{+1,0,5,15},{-1,0,7,13},{-1,4,1,15},{+1,4,3,13},{+1,12,1,7},{-1,12,3,5}},
    /* out 11 */ {{-1,0,5,11},{+1,0,7,9},{+1,4,1,11},{-1,4,3,9},{-1,8,1,7},{+1,8,3,5}},
    /* out 12 */ {{-1,4,9,14},{+1,4,10,13},{+1,8,5,14},{-1,8,6,13},{-1,12,5,10},{+1,12,6,9}},
    /* out 13 */ {{+1,0,9,14},{-1,0,10,13},{-1,8,1,14},{+1,8,2,13},{+1,12,1,10},{-1,12,2,9}},
    /* out 14 */ {{-1,0,5,14},{+1,0,6,13},{+1,4,1,14},{-1,4,2,13},{-1,12,1,6},{+1,12,2,5}},
    /* out 15 */ {{+1,0,5,10},{-1,0,6,9},{-1,4,1,10},{+1,4,2,9},{+1,8,1,6},{-1,8,2,5}},
};

void vho_invert4x4(const float e[16], float out[16])
{
    float inv[16];
    for (int o = 0; o < 16; ++o) {
        float acc = 0.0f;
        for (int k = 0; k < 6; ++k) {
            const signed char *c = k_cof[o][k];
            float t = (e[c[1]] * e[c[2]]) * e[c[3]];
            if (c[0] < 0) t = -t;
            acc = (k == 0) ? t : acc + t;
        }
        inv[o] = acc;
    }
    float det = e[0]*inv[0] + e[1]*inv[4] + e[2]*inv[8] + e[3]*inv[12];
    float detr = 1.0f / det;
    for (int i = 0; i < 16; ++i) out[i] = inv[i] * detr;
}

/* project, VoxelUtils.cu:770-777: M*p, three true divides by .z, implicit
 * float->int in make_int2. */
void vho_project(const float m[9], const float p[3], int32_t out[2])
{
    float q[3];
    mat3_mul_vec3(m, p, q);
    float qx = q[0] / q[2], qy = q[1] / q[2];
    out[0] = vho_float2int_rz(qx);
    out[1] = vho_float2int_rz(qy);
}

/* combineVoxel, VoxelUtils.cu:779-787 */
void vho_combine_voxel(const vho_voxel *o, const vho_voxel *c, float wmax, vho_voxel *out)
{
    vho_voxel n;
    n.sdf = ((o->sdf * o->weight) + (c->sdf * c->weight)) / (o->weight + c->weight);
    n.weight = fminf(wmax, o->weight + c->weight);
    *out = n;
}

/* Position of pixel (x,y) in the launch order of allocBlocksKernel's grid of
 * 16x16 tiles (VoxelUtils.cu:610-611,710-712): blocks x-fastest then y,
 * threads x-fastest then y. */
uint32_t vho_launch_rank(int x, int y, int width)
{
    uint32_t tilesX = (uint32_t)((width + 15) / 16);
    return (((uint32_t)(y >> 4) * tilesX + (uint32_t)(x >> 4)) << 8)
         + ((uint32_t)(y & 15) << 4) + (uint32_t)(x & 15);
}

/* block2World, VoxelUtils.cu:289-304: block min corner, voxel corner (no +0.5) */
static void block2world(const vho_table *t, const int32_t b[3], float out[3])
{
    for (int k = 0; k < 3; ++k) {
        int32_t v = wrap_mul(b[k], t->p.voxelBlockSize);
        out[k] = (float)v * t->p.voxelSize;
    }
}

/* blockInFrustum, VoxelUtils.cu:344-359.
 * REFERENCE: the block corner goes through global_transform (the author's own
 * TODO at :348 notes it should be the inverse) and the image bounds are the
 * only test.  PINHOLE: inverse pose, and the corner must be in front (z>0). */
int vho_block_in_frustum(const vho_table *t, const int32_t block[3])
{
    float w[4], c[4], q[3];
    block2world(t, block, w);
    w[3] = 1.0f;
    if (t->semantics == VHO_SEM_REFERENCE) {
        vho_mat4_mul_vec4(t->p.global_transform, w, c);
    } else {
        vho_mat4_mul_vec4(t->p.inv_global_transform, w, c);
        if (!(c[2] > 0.0f)) return 0;
    }
    mat3_mul_vec3(t->proj, c, q);
    float qx = q[0] / q[2], qy = q[1] / q[2];
    int32_t x = vho_float2int_rz(qx), y = vho_float2int_rz(qy);
    return (x < t->width && x >= 0 && y < t->height && y >= 0);
}

/* ------------------------------------------------------------------ */
/* lifecycle                                                           */
/* ------------------------------------------------------------------ */

/* common.h:39-50 as copied by SDF_Hashtable.cpp:62-73 */
void vho_default_params(vho_params *p)
{
    static const float I[16] = {1,0,0,0, 0,1,0,0, 0,0,1,0, 0,0,0,1};
    memset(p, 0, sizeof *p);
    memcpy(p->global_transform, I, sizeof I);
    memcpy(p->inv_global_transform, I, sizeof I);
    p->numBuckets = 5000;
    p->bucketSize = 5;
    p->attachedLinkedListSize = 4;
    p->numVoxelBlocks = 1000;
    p->voxelBlockSize = 8;
    p->voxelSize = 0.02f;
    p->numOccupiedBlocks = 0;
    p->maxIntegrationDistance = 4.0f;
    p->truncScale = 0.01f;
    p->truncation = 1.0f;
    p->integrationWeightSample = 10;
    p->integrationWeightMax = 255.0f;
}

static void reset_entries(vho_entry *e, size_t n)   /* resetHashTableKernel, :151-158 */
{
    /* (threads only to spread the first touch of multi-gigabyte tables; the values are per-entry constants) */
#pragma omp parallel for schedule(static) if (n > (1u << 20))
    for (size_t i = 0; i < n; ++i) {
        e[i].offset = 0;
        e[i].ptr = VHO_FREE_BLOCK;
        e[i].pos[0] = e[i].pos[1] = e[i].pos[2] = VHO_POS_SENTINEL;
    }
}

/* deviceAllocate + calculateKinectProjectionMatrix, VoxelUtils.cu:169-231.
 * The compact table and the volume (GL buffers in the reference,
 * SDFRenderer.cpp:34-61) are owned here and zero-initialised. */
vho_table *vho_create(const vho_params *p, int width, int height, int semantics)
{
    return vho_create_shard(p, width, height, semantics, 0, p->numBuckets);
}

/* A shard owns the buckets [lo, hi) of a logical table of numBuckets buckets
 * (build extension for multi-GPU); keys that hash outside
 * the range are ignored by insert / lookup. */
vho_table *vho_create_shard(const vho_params *p, int width, int height, int semantics,
                            uint32_t lo, uint32_t hi)
{
    if (p->voxelBlockSize != 8 || p->numBuckets == 0 || p->bucketSize == 0) return NULL;
    if (lo >= hi || hi > p->numBuckets) return NULL;
    vho_table *t = (vho_table *)calloc(1, sizeof *t);
    if (!t) return NULL;
    t->p = *p;
    t->width = width;
    t->height = height;
    t->semantics = semantics;
    t->bucket_lo = lo;
    t->bucket_hi = hi;
    size_t n = (size_t)(hi - lo) * p->bucketSize;
    t->heap    = (uint32_t *)malloc(sizeof(uint32_t) * p->numVoxelBlocks);
    t->table   = (vho_entry *)malloc(sizeof(vho_entry) * n);
    t->compact = (vho_entry *)malloc(sizeof(vho_entry) * n);
    t->mutex   = (int32_t *)calloc(hi - lo, sizeof(int32_t));
    t->slot_epoch = (uint32_t *)calloc(n, sizeof(uint32_t));
    t->blocks  = (vho_voxel *)calloc((size_t)p->numVoxelBlocks * 512, sizeof(vho_voxel));
    if (!t->heap || !t->table || !t->compact || !t->mutex || !t->blocks || !t->slot_epoch) {
        vho_destroy(t);
        return NULL;
    }
    reset_entries(t->table, n);
    reset_entries(t->compact, n);
    for (uint32_t i = 0; i < p->numVoxelBlocks; ++i) t->heap[i] = i;   /* resetHeapKernel */
    t->heap_counter = (int32_t)p->numVoxelBlocks - 1;                    /* :207 */
    t->compact_counter = 0;

    /* common.h:7-10,15-16 scaled with the resolution */
    const float sx = (float)width / 640.0f, sy = (float)height / 480.0f;
    const float fx = 517.3f * sx, fy = 516.5f * sy, cx = 318.6f * sx, cy = 255.3f * sy;
    const float KT[9] = {fx, 0, 0,  0, fy, 0,  cx, cy, 1};   /* intrinsicsTranspose read row-major */
    const float K[9]  = {fx, 0, cx, 0, fy, cy, 0, 0, 1};
    memcpy(t->proj, semantics == VHO_SEM_REFERENCE ? KT : K, sizeof K);
    t->rc_fx = fx; t->rc_fy = fy; t->rc_cx = cx; t->rc_cy = cy;
    return t;
}

void vho_destroy(vho_table *t)
{
    if (!t) return;
    free(t->heap); free(t->table); free(t->compact); free(t->mutex); free(t->blocks); free(t->slot_epoch);
    free(t);
}

void vho_set_projection(vho_table *t, const float m[9]) { memcpy(t->proj, m, 9 * sizeof(float)); }

void vho_set_raycast_intrinsics(vho_table *t, float fx, float fy, float cx, float cy)
{
    t->rc_fx = fx; t->rc_fy = fy; t->rc_cx = cx; t->rc_cy = cy;
}

/* ------------------------------------------------------------------ */
/* per-frame steps                                                     */
/* ------------------------------------------------------------------ */

/* SDF_Hashtable.cpp:15-18 */
void vho_set_pose(vho_table *t, const float pose[16])
{
    float inv[16];
    vho_invert4x4(pose, inv);
    memcpy(t->p.global_transform, pose, sizeof inv);
    memcpy(t->p.inv_global_transform, inv, sizeof inv);
}

/* resetHashTableMutexes, VoxelUtils.cu:146-149 */
void vho_reset_mutexes(vho_table *t)
{
    memset(t->mutex, 0, sizeof(int32_t) * (t->bucket_hi - t->bucket_lo));
    t->epoch += 1;
}

/* allocSingleBlockInHeap, VoxelUtils.cu:328-334, with the exhaustion case
 * defined (the reference reads heap[-1]): an empty heap refuses the request
 * and leaves the counter alone. */
static int32_t heap_pop(vho_table *t)
{
    if (t->heap_counter < 0) return -1;
    int32_t addr = t->heap_counter--;
    return (int32_t)t->heap[addr];
}

/* insertVoxelEntry, live part VoxelUtils.cu:421-456 */
static void insert_entry_overflow(vho_table *t, const int32_t key[3]);

static void insert_entry(vho_table *t, const int32_t key[3])
{
    if (t->overflow) { insert_entry_overflow(t, key); return; }
    const uint32_t bs = t->p.bucketSize, nb = t->p.numBuckets;
    const uint32_t hg = vho_hash(key[0], key[1], key[2], nb);
    if (hg < t->bucket_lo || hg >= t->bucket_hi) return;               /* another shard's bucket */
    const uint32_t h = hg - t->bucket_lo;
    const uint32_t start = h * bs;
    int saw_free = 0;
    for (uint32_t i = 0; i < bs; ++i) {
        uint32_t idx = start + i;   /* (start+i) % N in the reference; never wraps */
        vho_entry *e = &t->table[idx];
        if (e->pos[0] == key[0] && e->pos[1] == key[1] && e->pos[2] == key[2]
            && e->ptr != VHO_FREE_BLOCK) return;                       /* already there */
        if (e->ptr == VHO_FREE_BLOCK) {
            saw_free = 1;
            int32_t prev = t->mutex[h];                                /* atomicExch */
            t->mutex[h] = VHO_LOCKED_BLOCK;
            if (prev != VHO_LOCKED_BLOCK) {
                int32_t blk = heap_pop(t);
                if (blk < 0) { t->stats.heap_exhausted++; return; }
                e->pos[0] = key[0]; e->pos[1] = key[1]; e->pos[2] = key[2];
                e->offset = 0;
                e->ptr = blk * 512;
                t->stats.inserted++;
                return;
            }
            /* bucket already locked this frame: the scan goes on and every
             * later free slot loses the exchange again */
        }
    }
    if (saw_free) t->stats.lock_losses++; else t->stats.bucket_full++;
}

/* ------------------------------------------------------------------ */
/* overflow linked list (opt-in)                */
/* ------------------------------------------------------------------ */
/* The reference carries the idea as dead, unfinished code (#ifdef LINKED_LIST_ENABLED, never defined:
 * lookup tail VoxelUtils.cu:384-411, insert tail :458-539, delete tail :578-602, beforeThis() missing).
 * Its lookup tail is coherent and is restated verbatim; insert and delete are completed the way the
 * paper the demo follows does it (Niessner et al. 2013, 3.1 / 4.2), keeping what the dead code shows
 * of the author's intent:
 *   - a key whose home bucket is full lives in a free slot of a following bucket; the entries of one
 *     home bucket form a chain that starts in the bucket's LAST slot and is linked through `offset`,
 *     measured from that last slot, modulo the table (:388-399: i = lastEntryInBucket + curr.offset;
 *     i %= numBuckets*bucketSize);
 *   - lookups follow at most attachedLinkedListSize iterations of that loop (:391-392), i.e. the last
 *     slot plus attachedLinkedListSize-1 chained entries; insertion refuses to grow a chain beyond what
 *     the loop can reach;
 *   - the free slot is searched among the 9 slots behind the bucket's last slot (:475-478 looks ahead
 *     `j < 10`), never in a bucket's last slot (it is that bucket's own chain head);
 *   - the home bucket is locked first, then the bucket that holds the free slot (:472-482); both stay
 *     locked for the frame, so "at most one insertion per bucket per frame" holds for both;
 *   - the new entry goes to the FRONT of the chain (entry.offset = head.offset; head.offset = j): only
 *     the two locked buckets are written.
 * A slot that received its entry earlier in the SAME lock epoch counts as free in the look-ahead (its
 * bucket is locked, so the insertion fails): contenders are then judged against the table as it was
 * when the epoch began, whatever order cameras are served in -- the rule the GPU's single claim launch
 * realises.
 * With the list on, a bucket's entries no longer form a prefix of its slots (deletion leaves holes,
 * as in the paper), so every scan covers all bucketSize slots. */
void vho_set_overflow(vho_table *t, int enabled, uint32_t segment_buckets)
{
    t->overflow = enabled != 0;
    t->overflow_seg = segment_buckets;
}

/* chain arithmetic: entries [base, base+n) of the segment that holds local bucket h */
static void chain_segment(const vho_table *t, uint32_t h, size_t *base, size_t *n)
{
    const uint32_t bs = t->p.bucketSize, owned = t->bucket_hi - t->bucket_lo;
    if (t->overflow_seg == 0 || t->overflow_seg >= t->p.numBuckets) { *base = 0; *n = (size_t)owned * bs; return; }
    const uint32_t hg = h + t->bucket_lo;
    uint32_t lo = (hg / t->overflow_seg) * t->overflow_seg, hi = lo + t->overflow_seg;
    if (hi > t->p.numBuckets) hi = t->p.numBuckets;
    if (lo < t->bucket_lo) lo = t->bucket_lo;
    if (hi > t->bucket_hi) hi = t->bucket_hi;
    *base = (size_t)(lo - t->bucket_lo) * bs;
    *n = (size_t)(hi - lo) * bs;
}

static size_t chain_slot(size_t last, int32_t offset, size_t base, size_t n)
{
    return base + (size_t)(((int64_t)(last - base) + (int64_t)offset) % (int64_t)n);
}

static int key_at(const vho_entry *e, const int32_t key[3])
{
    return e->pos[0] == key[0] && e->pos[1] == key[1] && e->pos[2] == key[2] && e->ptr != VHO_FREE_BLOCK;
}

/* getVoxelEntry4Block with the list (:362-411): entry index or -1; *prev_out = chain predecessor
 * (entry index) when the key was found behind the bucket's last slot, else -1 */
static int64_t find_overflow(const vho_table *t, const int32_t key[3], uint32_t h, int64_t *prev_out)
{
    const uint32_t bs = t->p.bucketSize, L = t->p.attachedLinkedListSize;
    const size_t start = (size_t)h * bs, last = start + bs - 1;
    if (prev_out) *prev_out = -1;
    for (uint32_t i = 0; i < bs; ++i)
        if (key_at(&t->table[start + i], key)) return (int64_t)(start + i);            /* :374-381 */
    size_t base, n;
    chain_segment(t, h, &base, &n);
    size_t i = last, prev = last;
    for (uint32_t iter = 0; iter < L; ++iter) {                                          /* :391-392 */
        const vho_entry *curr = &t->table[i];
        if (key_at(curr, key)) { if (prev_out && i != last) *prev_out = (int64_t)prev; return (int64_t)i; }
        if (curr->offset == 0) break;                                                    /* :396 */
        prev = i;
        i = chain_slot(last, curr->offset, base, n);                                     /* :398-399 */
    }
    return -1;
}

static void insert_entry_overflow(vho_table *t, const int32_t key[3])
{
    const uint32_t bs = t->p.bucketSize, nb = t->p.numBuckets, L = t->p.attachedLinkedListSize;
    const uint32_t hg = vho_hash(key[0], key[1], key[2], nb);
    if (hg < t->bucket_lo || hg >= t->bucket_hi) return;
    const uint32_t h = hg - t->bucket_lo;
    const size_t start = (size_t)h * bs, last = start + bs - 1;
    int64_t first_empty = -1;
    for (uint32_t i = 0; i < bs; ++i) {
        const vho_entry *e = &t->table[start + i];
        if (key_at(e, key)) return;                                                      /* already there */
        if (first_empty < 0 && e->ptr == VHO_FREE_BLOCK) first_empty = (int64_t)(start + i);
    }
    size_t base, n;
    chain_segment(t, h, &base, &n);
    /* the lookup loop again: presence, and how long the chain is */
    uint32_t links = 0;
    int ended = 0;
    size_t i = last;
    for (uint32_t iter = 0; iter < L; ++iter) {
        const vho_entry *curr = &t->table[i];
        if (key_at(curr, key)) return;
        if (curr->offset == 0) { ended = 1; break; }
        i = chain_slot(last, curr->offset, base, n);
        ++links;
    }
    if (first_empty >= 0) {                                        /* a slot of the home bucket: as without the list */
        const int32_t prev = t->mutex[h];
        t->mutex[h] = VHO_LOCKED_BLOCK;
        if (prev == VHO_LOCKED_BLOCK) { t->stats.lock_losses++; return; }
        const int32_t blk = heap_pop(t);
        if (blk < 0) { t->stats.heap_exhausted++; return; }
        vho_entry *e = &t->table[first_empty];
        e->pos[0] = key[0]; e->pos[1] = key[1]; e->pos[2] = key[2];
        e->ptr = blk * 512;                                        /* (a free last slot has offset 0: no chain without a head) */
        t->slot_epoch[first_empty] = t->epoch;
        t->stats.inserted++;
        return;
    }
    if (!ended || links + 1 > L - 1 || L < 2) { t->stats.bucket_full++; return; }   /* chain at the reach of the lookup loop */
    int64_t target = -1;
    int32_t tj = 0;
    for (int32_t j = 1; j < 10; ++j) {                                               /* :475-478 */
        const size_t s = chain_slot(last, j, base, n);
        if (s % bs == bs - 1) continue;                            /* another bucket's chain head */
        if (t->table[s].ptr == VHO_FREE_BLOCK || t->slot_epoch[s] == t->epoch) { target = (int64_t)s; tj = j; break; }
    }
    if (target < 0) { t->stats.bucket_full++; return; }
    int32_t prev = t->mutex[h];                                                      /* [1] lock the parent block, :472 */
    t->mutex[h] = VHO_LOCKED_BLOCK;
    if (prev == VHO_LOCKED_BLOCK) { t->stats.lock_losses++; return; }
    const uint32_t hb = (uint32_t)((size_t)target / bs);
    prev = t->mutex[hb];                                                             /* [3] now lock this new bucket, :482 */
    t->mutex[hb] = VHO_LOCKED_BLOCK;
    if (prev == VHO_LOCKED_BLOCK) { t->stats.lock_losses++; return; }
    const int32_t blk = heap_pop(t);
    if (blk < 0) { t->stats.heap_exhausted++; return; }
    vho_entry *e = &t->table[target];
    e->pos[0] = key[0]; e->pos[1] = key[1]; e->pos[2] = key[2];
    e->ptr = blk * 512;
    e->offset = t->table[last].offset;
    t->table[last].offset = tj;
    t->slot_epoch[target] = t->epoch;
    t->stats.inserted++;
}

/* Truncation-band allocation (opt-in extension; the reference has the
 * idea commented out, VoxelUtils.cu:632-703).  With alloc_band = 0 a pixel demands the block of
 * its surface point only (the live reference behaviour).  With alloc_band = b > 0 it demands the
 * blocks of 2*ceil(b/step)+1 points on its viewing ray at camera depths z + (k - half)*step,
 * step = half a block edge; the middle sample is the surface point itself, bit for bit. */
#define VHO_MAX_BAND_SAMPLES 64

void vho_set_alloc_band(vho_table *t, float band) { t->alloc_band = band > 0.0f ? band : 0.0f; }

static int band_samples(const vho_table *t, float *step_out)
{
    const float step = 4.0f * t->p.voxelSize;
    *step_out = step;
    if (!(t->alloc_band > 0.0f)) return 1;
    int half = (int)ceilf(t->alloc_band / step);
    if (half > (VHO_MAX_BAND_SAMPLES - 1) / 2) half = (VHO_MAX_BAND_SAMPLES - 1) / 2;
    return 2 * half + 1;
}

/* block key of sample k of the pixel with camera-space vertex v; 0 if there is none */
static int band_key(const vho_table *t, const float *v, int k, int nS, float step, int32_t key[3])
{
    const int half = (nS - 1) / 2;
    const float s = v[2] + ((float)k - (float)half) * step;
    if (k != half && !(s > 0.0f)) return 0;      /* the surface sample is never filtered (:621 only tests z != 0) */
    const float scale = s / v[2];
    const float p[4] = { v[0] * scale, v[1] * scale, s, v[3] };
    const float *src = (k == half) ? v : p;                           /* the surface sample is v itself */
    float g[4];
    vho_mat4_mul_vec4(t->p.global_transform, src, g);                 /* :622, w as stored */
    vho_world2block(g, t->p.voxelSize, t->p.voxelBlockSize, key);     /* :636 */
    return 1;
}

/* Second band mode, VHO_BAND_NORMAL_DDA: what the reference has commented out in allocBlocksKernel
 * (VoxelUtils.cu:632-633 the two ray ends p -+ truncation*n, :641-656 step / tMax / tDelta, :678-699 the
 * walk): every block the segment from p - b*n to p + b*n crosses is demanded, visited by a block DDA
 * (Amanatides-Woo).  p = the pixel's world point (:622), n = its normal from preProcess's normal map
 * rotated into the world frame, b = alloc_band.  A pixel without a normal (preProcess writes 0 at
 * the border and next to invalid depth) demands its surface block only.  Restated so that oracle and
 * kernel can agree bit for bit: the segment is parametrised over [0,1] (no normalisation, no square
 * root); block k covers world coordinates [(8k - 0.5) * voxelSize, (8k + 7.5) * voxelSize) on an
 * axis, which is what world2Block's rounding (:280-287, :266-278) maps to k; ties are broken as in
 * :683-698 (x only if strictly smallest, then z if smaller than y, else y); the walk ends at the
 * end block, after 62 steps (rank bits), or when the next crossing lies beyond the segment's end. */
void vho_set_band_mode(vho_table *t, int mode) { t->band_mode = mode; }
void vho_set_normals(vho_table *t, const float *normals) { t->normals = normals; }

/* the block DDA of both DDA band modes: every block the segment start -> end crosses, from start's block on */
static int dda_walk(const vho_table *t, const float start[3], const float end[3], int32_t keys[][3])
{
    const float vs = t->p.voxelSize;
    float dir[3];
    for (int a = 0; a < 3; ++a) dir[a] = end[a] - start[a];
    int32_t cur[3], last[3];
    vho_world2block(start, vs, t->p.voxelBlockSize, cur);
    vho_world2block(end, vs, t->p.voxelBlockSize, last);
    int32_t step[3];
    float tmax[3], tdelta[3];
    for (int a = 0; a < 3; ++a) {
        step[a] = dir[a] > 0.0f ? 1 : dir[a] < 0.0f ? -1 : 0;
        if (step[a] == 0) { tmax[a] = VHO_INF_F; tdelta[a] = VHO_INF_F; continue; }   /* :658-668 */
        const float boundary = ((float)wrap_mul(wrap_add(cur[a], step[a] > 0 ? 1 : 0), 8) - 0.5f) * vs;
        tmax[a] = (boundary - start[a]) / dir[a];
        tdelta[a] = (8.0f * vs) / fabsf(dir[a]);
    }
    int n = 0;
    keys[n][0] = cur[0]; keys[n][1] = cur[1]; keys[n][2] = cur[2];
    ++n;
    while ((cur[0] != last[0] || cur[1] != last[1] || cur[2] != last[2]) && n < VHO_MAX_BAND_SAMPLES - 1) {
        int a;
        if (tmax[0] < tmax[1] && tmax[0] < tmax[2]) a = 0;            /* :683 */
        else if (tmax[2] < tmax[1]) a = 2;                            /* :688 */
        else a = 1;                                                   /* :693 */
        if (!(tmax[a] <= 1.0f)) break;
        cur[a] = wrap_add(cur[a], step[a]);
        tmax[a] += tdelta[a];
        keys[n][0] = cur[0]; keys[n][1] = cur[1]; keys[n][2] = cur[2];
        ++n;
    }
    return n;
}

static int dda_keys(const vho_table *t, const float *v, const float *nrm, int32_t keys[][3])
{
    float g[4];
    vho_mat4_mul_vec4(t->p.global_transform, v, g);                   /* :622, w as stored */
    vho_world2block(g, t->p.voxelSize, t->p.voxelBlockSize, keys[0]);
    if (!nrm || (nrm[0] == 0.0f && nrm[1] == 0.0f && nrm[2] == 0.0f) || nrm[0] != nrm[0] || nrm[1] != nrm[1] ||
        nrm[2] != nrm[2])
        return 1;
    const float *T = t->p.global_transform;
    const float b = t->alloc_band;
    float start[3], end[3];
    for (int a = 0; a < 3; ++a) {
        const float nw = T[4*a+0]*nrm[0] + T[4*a+1]*nrm[1] + T[4*a+2]*nrm[2];
        start[a] = g[a] - (b * nw);                                   /* :632 */
        end[a] = g[a] + (b * nw);                                     /* :633 */
    }
    return dda_walk(t, start, end, keys);
}

/* Third band mode, VHO_BAND_RAY_DDA (round 4): the same block DDA along the pixel's VIEWING RAY instead of its normal --
 * every block the world-space segment from the ray's point at camera depth z - b to its point at depth z + b crosses
 * (z = the vertex's depth, b = alloc_band; a band that would begin at or behind the camera begins at the surface point
 * instead).  What the ray samples of VHO_BAND_RAY approximate with 2*ceil(b / half a block)+1 points: here the blocks
 * are exactly those the segment crosses, one transform per segment end and the DDA's divisions per PIXEL, nothing per
 * sample.  The two ends: the vertex scaled to the end's depth, (x*s/z, y*s/z, s, w), through global_transform as :622. */
static int ray_dda_keys(const vho_table *t, const float *v, int32_t keys[][3])
{
    const float z = v[2], b = t->alloc_band;
    float s0 = z - b;
    if (!(s0 > 0.0f)) s0 = z;
    const float s1 = z + b;
    const float c0 = s0 / z, c1 = s1 / z;
    const float p0[4] = { v[0] * c0, v[1] * c0, s0, v[3] }, p1[4] = { v[0] * c1, v[1] * c1, s1, v[3] };
    float g0[4], g1[4];
    vho_mat4_mul_vec4(t->p.global_transform, p0, g0);
    vho_mat4_mul_vec4(t->p.global_transform, p1, g1);
    return dda_walk(t, g0, g1, keys);
}

/* the block keys pixel (x,y) demands, in rank order; valid[k] = 0 where a ray sample has no key */
static int pixel_keys(const vho_table *t, const float *verts, int x, int y, int nS, float step, int32_t keys[][3],
                      uint8_t *valid)
{
    const float *v = verts + 4 * ((size_t)y * t->width + x);
    if (t->band_mode == VHO_BAND_NORMAL_DDA && t->alloc_band > 0.0f) {
        const int n = dda_keys(t, v, t->normals ? t->normals + 4 * ((size_t)y * t->width + x) : NULL, keys);
        for (int k = 0; k < n; ++k) valid[k] = 1;
        return n;
    }
    if (t->band_mode == VHO_BAND_RAY_DDA && t->alloc_band > 0.0f) {
        const int n = ray_dda_keys(t, v, keys);
        for (int k = 0; k < n; ++k) valid[k] = 1;
        return n;
    }
    for (int k = 0; k < nS; ++k) valid[k] = (uint8_t)band_key(t, v, k, nS, step, keys[k]);
    return nS;
}

/* allocBlocksKernel, VoxelUtils.cu:606-705, visited in launch order */
void vho_alloc_blocks(vho_table *t, const float *verts)
{
    const int W = t->width, H = t->height;
    const int tilesX = (W + 15) / 16, tilesY = (H + 15) / 16;
    float step;
    const int nS = band_samples(t, &step);
    for (int by = 0; by < tilesY; ++by)
    for (int bx = 0; bx < tilesX; ++bx)
    for (int ty = 0; ty < 16; ++ty)
    for (int tx = 0; tx < 16; ++tx) {
        const int x = bx * 16 + tx, y = by * 16 + ty;
        if (x >= W || y >= H) continue;
        const float *v = verts + 4 * ((size_t)y * W + x);
        if (v[2] == 0.0f) continue;                                   /* :621 */
        t->stats.pixels_valid++;
        int counted = 0;
        int32_t keys[VHO_MAX_BAND_SAMPLES][3];
        uint8_t valid[VHO_MAX_BAND_SAMPLES];
        const int n = pixel_keys(t, verts, x, y, nS, step, keys, valid);
        for (int k = 0; k < n; ++k) {
            if (!valid[k]) continue;
            if (!vho_block_in_frustum(t, keys[k])) continue;          /* :673 */
            if (!counted) { t->stats.pixels_in_frustum++; counted = 1; }
            insert_entry(t, keys[k]);
        }
    }
}

/* flattenKernel / flattenIntoBuffer, VoxelUtils.cu:719-768.  The order of the
 * compact list is a race in the reference; here it is table order.  The
 * redundant reset of the whole compact table (:757-758) is not repeated. */
int vho_flatten(vho_table *t)
{
    const size_t n = (size_t)(t->bucket_hi - t->bucket_lo) * t->p.bucketSize;
    int count = 0;
    for (size_t i = 0; i < n; ++i) {
        const vho_entry *e = &t->table[i];
        if (e->ptr != VHO_FREE_BLOCK && vho_block_in_frustum(t, e->pos))
            t->compact[count++] = *e;
    }
    t->compact_counter = count;
    t->p.numOccupiedBlocks = (uint32_t)count;
    t->stats.occupied = (uint32_t)count;
    return count;
}

/* integrateDepthMapKernel, VoxelUtils.cu:790-842 */
static void integrate_depth(vho_table *t, const float *depth_base, int stride);

void vho_integrate_depth_map(vho_table *t, const float *verts)
{
    integrate_depth(t, verts + 2, 4);          /* verts[idx].z */
}

/* one voxel of integrateDepthMapKernel (:793-840); returns 1 if the voxel was written */
static int integrate_voxel(vho_table *t, const vho_entry *e, const int32_t base[3], int tx, int ty, int tz,
                           const float *depth_base, int stride)
{
    const int W = t->width, H = t->height;
    const int32_t vi[3] = { wrap_add(base[0], tx), wrap_add(base[1], ty), wrap_add(base[2], tz) };
    float pc[3];   /* camera-space point handed to project() */
    if (t->semantics == VHO_SEM_REFERENCE) {
        /* :797-800 -- the inverse pose is applied to the voxel INDEX, the
         * result is truncated back to an int index, then scaled to metres */
        const float vf[4] = { (float)vi[0], (float)vi[1], (float)vi[2], 1.0f };
        float r[4];
        vho_mat4_mul_vec4(t->p.inv_global_transform, vf, r);
        for (int k = 0; k < 3; ++k)
            pc[k] = (float)vho_float2int_rz(r[k]) * t->p.voxelSize;
    } else {
        const float wv[4] = { (float)vi[0] * t->p.voxelSize, (float)vi[1] * t->p.voxelSize,
                              (float)vi[2] * t->p.voxelSize, 1.0f };
        float r[4];
        vho_mat4_mul_vec4(t->p.inv_global_transform, wv, r);
        pc[0] = r[0]; pc[1] = r[1]; pc[2] = r[2];
    }
    int32_t s[2];
    vho_project(t->proj, pc, s);                                          /* :801 */
    if (s[0] < 0 || s[0] >= W || s[1] < 0 || s[1] >= H) return 0;         /* :803 */
    const float depth = depth_base[(size_t)stride * ((size_t)s[1] * W + s[0])];  /* :805 */
    if (depth <= 0) return 0;                                             /* :806 */
    float sdf = depth - pc[2];                                            /* :813 */
    float trunc = t->p.truncation;                                        /* :815 */
    if (t->integrate_flags & VHO_INT_DEPTH_TRUNCATION)                    /* the commented half of :815 = getTruncation, :261-264 */
        trunc = t->p.truncation + (t->p.truncScale * depth);
    if (!(sdf > -trunc)) return 0;                                        /* :818 */
    sdf = (sdf >= 0) ? fminf(trunc, sdf) : fmaxf(-trunc, sdf);            /* :819-824 */
    float weight = 0.1f;                                                  /* :829 */
    if (t->integrate_flags & VHO_INT_WEIGHT_SAMPLE) {                     /* the commented :827 with :808-811 */
        const float range_min = 0.5f, range_max = 5.0f;
        const float zero_one = (depth - range_min) / (range_max - range_min);
        weight = fmaxf((float)((double)t->p.integrationWeightSample * 1.5 * (1.0 - (double)zero_one)), 1.0f);
    }
    const vho_voxel cur = { sdf, weight };
    vho_voxel *dst = &t->blocks[(size_t)e->ptr + (size_t)(tz * 64 + ty * 8 + tx)];  /* :836 */
    vho_combine_voxel(dst, &cur, t->p.integrationWeightMax, dst);
    return 1;
}

static void integrate_depth(vho_table *t, const float *depth_base, int stride)
{
    for (int b = 0; b < t->compact_counter; ++b) {
        const vho_entry *e = &t->compact[b];
        int32_t base[3];
        for (int k = 0; k < 3; ++k) base[k] = wrap_mul(e->pos[k], t->p.voxelBlockSize);  /* :793 */
        for (int tz = 0; tz < 8; ++tz)
        for (int ty = 0; ty < 8; ++ty)
        for (int tx = 0; tx < 8; ++tx)
            t->stats.voxels_updated += (uint32_t)integrate_voxel(t, e, base, tx, ty, tz, depth_base, stride);
    }
}

void vho_set_integrate_flags(vho_table *t, int flags) { t->integrate_flags = flags; }

/* SDF_Hashtable::integrate, SDF_Hashtable.cpp:11-40 */
int vho_integrate(vho_table *t, const float pose[16], const float *verts, vho_frame_stats *stats)
{
    memset(&t->stats, 0, sizeof t->stats);
    vho_set_pose(t, pose);
    vho_reset_mutexes(t);
    vho_alloc_blocks(t, verts);
    int occ = vho_flatten(t);
    if (occ > 0) vho_integrate_depth_map(t, verts);    /* :848 */
    t->stats.heap_counter = t->heap_counter;
    if (stats) *stats = t->stats;
    return occ;
}

/* ------------------------------------------------------------------ */
/* the same frame on several host threads (bench.py's cpu_baseline;     */
/* SURVEY.md 8(d) "all cores via a static split")                       */
/* ------------------------------------------------------------------ */
/* Identical results to vho_integrate: the per-pixel key computation (transform, rounding, frustum
 * test) runs in parallel into a key array, the insertions stay serial in launch order (H8 is a
 * sequential contract); the table walk is cut into one chunk per thread whose hits are concatenated
 * in chunk order (= table order); the blocks of the compact list are independent. */
#include <omp.h>

int vho_integrate_mt(vho_table *t, const float pose[16], const float *verts, int threads, vho_frame_stats *stats)
{
    if (threads < 1) threads = 1;
    memset(&t->stats, 0, sizeof t->stats);
    vho_set_pose(t, pose);
    vho_reset_mutexes(t);
    const int W = t->width, H = t->height;
    float step;
    const int nS = band_samples(t, &step);
    /* ---- allocBlocks: keys in parallel, insertions in launch order ---- */
    const int maxK = (t->band_mode != VHO_BAND_RAY && t->alloc_band > 0.0f) ? VHO_MAX_BAND_SAMPLES : nS;
    int32_t *keys = (int32_t *)malloc((size_t)W * H * maxK * 4 * sizeof(int32_t));   /* {x,y,z,wanted} */
    if (!keys) return -1;
    #pragma omp parallel for num_threads(threads) schedule(static)
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const float *v = verts + 4 * ((size_t)y * W + x);
            int32_t *kk = keys + 4 * (((size_t)y * W + x) * maxK);
            for (int k = 0; k < maxK; ++k) kk[4 * k + 3] = 0;
            if (v[2] == 0.0f) continue;
            int32_t pk[VHO_MAX_BAND_SAMPLES][3];
            uint8_t valid[VHO_MAX_BAND_SAMPLES];
            const int n = pixel_keys(t, verts, x, y, nS, step, pk, valid);
            for (int k = 0; k < n; ++k) {
                if (!valid[k]) continue;
                kk[4 * k + 0] = pk[k][0]; kk[4 * k + 1] = pk[k][1]; kk[4 * k + 2] = pk[k][2];
                kk[4 * k + 3] = vho_block_in_frustum(t, pk[k]) ? 1 : 0;
            }
        }
    const int tilesX = (W + 15) / 16, tilesY = (H + 15) / 16;
    for (int by = 0; by < tilesY; ++by)
    for (int bx = 0; bx < tilesX; ++bx)
    for (int ty = 0; ty < 16; ++ty)
    for (int tx = 0; tx < 16; ++tx) {
        const int x = bx * 16 + tx, y = by * 16 + ty;
        if (x >= W || y >= H) continue;
        if (verts[4 * ((size_t)y * W + x) + 2] == 0.0f) continue;
        t->stats.pixels_valid++;
        int counted = 0;
        for (int k = 0; k < maxK; ++k) {
            const int32_t *kk = keys + 4 * (((size_t)y * W + x) * maxK + k);
            if (!kk[3]) continue;
            if (!counted) { t->stats.pixels_in_frustum++; counted = 1; }
            insert_entry(t, kk);
        }
    }
    free(keys);
    /* ---- flatten: one chunk per thread, concatenated in chunk order ---- */
    const size_t n = (size_t)(t->bucket_hi - t->bucket_lo) * t->p.bucketSize;
    size_t *cnt = (size_t *)calloc((size_t)threads + 1, sizeof(size_t));
    uint32_t **found = (uint32_t **)calloc((size_t)threads, sizeof(uint32_t *));
    #pragma omp parallel num_threads(threads)
    {
        const int me = omp_get_thread_num(), nt = omp_get_num_threads();
        const size_t lo = n * (size_t)me / (size_t)nt, hi = n * (size_t)(me + 1) / (size_t)nt;
        uint32_t *mine = NULL;
        size_t m = 0, cap = 0;
        for (size_t i = lo; i < hi; ++i) {
            const vho_entry *e = &t->table[i];
            if (e->ptr == VHO_FREE_BLOCK || !vho_block_in_frustum(t, e->pos)) continue;
            if (m == cap) { cap = cap ? 2 * cap : 256; mine = (uint32_t *)realloc(mine, cap * sizeof(uint32_t)); }
            mine[m++] = (uint32_t)i;
        }
        found[me] = mine;
        cnt[me + 1] = m;
    }
    for (int i = 0; i < threads; ++i) cnt[i + 1] += cnt[i];
    #pragma omp parallel for num_threads(threads) schedule(static, 1)
    for (int i = 0; i < threads; ++i)
        for (size_t j = 0; j < cnt[i + 1] - cnt[i]; ++j) t->compact[cnt[i] + j] = t->table[found[i][j]];
    const int count = (int)cnt[threads];
    for (int i = 0; i < threads; ++i) free(found[i]);
    free(found);
    free(cnt);
    t->compact_counter = count;
    t->p.numOccupiedBlocks = (uint32_t)count;
    t->stats.occupied = (uint32_t)count;
    /* ---- integrateDepthMap: blocks are independent ---- */
    if (count > 0) {
        const float *depth_base = verts + 2;
        uint32_t updated = 0;
        #pragma omp parallel for num_threads(threads) schedule(dynamic, 4) reduction(+:updated)
        for (int b = 0; b < count; ++b) {
            const vho_entry *e = &t->compact[b];
            int32_t base[3];
            for (int k = 0; k < 3; ++k) base[k] = wrap_mul(e->pos[k], t->p.voxelBlockSize);
            for (int tz = 0; tz < 8; ++tz)
            for (int ty = 0; ty < 8; ++ty)
            for (int tx = 0; tx < 8; ++tx)
                updated += (uint32_t)integrate_voxel(t, e, base, tx, ty, tz, depth_base, 4);
        }
        t->stats.voxels_updated = updated;
    }
    t->stats.heap_counter = t->heap_counter;
    if (stats) *stats = t->stats;
    return count;
}

/* ------------------------------------------------------------------ */
/* raycast (build spec; the reference's pass is disabled and broken,    */
/* SDFRenderer.cpp:215-254, raycastSDF.frag:121-177)                   */
/* ------------------------------------------------------------------ */

/* getVoxelEntry4Block, live half VoxelUtils.cu:362-382: bucket scan, returns
 * the entry index or -1. */
static int64_t lookup_block(const vho_table *t, const int32_t key[3])
{
    const uint32_t bs = t->p.bucketSize;
    const uint32_t hg = vho_hash(key[0], key[1], key[2], t->p.numBuckets);
    if (hg < t->bucket_lo || hg >= t->bucket_hi) return -1;
    const uint32_t h = hg - t->bucket_lo;
    if (t->overflow) return find_overflow(t, key, h, NULL);
    for (uint32_t i = 0; i < bs; ++i) {
        const vho_entry *e = &t->table[(size_t)h * bs + i];
        if (e->pos[0] == key[0] && e->pos[1] == key[1] && e->pos[2] == key[2]
            && e->ptr != VHO_FREE_BLOCK) return (int64_t)h * bs + i;
    }
    return -1;
}

/* One ray per pixel from pose (camera->world) through the pinhole
 * (fx,fy,cx,cy).  Samples sit at camera depth t_i = t_min + i*voxelSize;
 * every sample is classified from the NEAREST voxel (world2Voxel, as the
 * reference's shader samples un-interpolated, raycastSDF.frag:101-105).  A
 * sample is valid when its block is in the table and its voxel has weight>0.
 * The surface is the first pair of consecutive valid samples with
 * sdf_prev > 0 >= sdf_cur; the reported camera depth interpolates linearly
 * between them.  0 = no hit. */
void vho_raycast(vho_table *t, const float pose[16], float t_min, float t_max, float *depth_out)
{
    const int W = t->width, H = t->height;
    const float dt = t->p.voxelSize;
    const int nsteps = vho_float2int_rz((t_max - t_min) / dt) + 1;
    /* rays are independent and only read the model: rows are spread over the host's threads */
#pragma omp parallel for schedule(dynamic, 4)
    for (int v = 0; v < H; ++v)
    for (int u = 0; u < W; ++u) {
        const float dx = ((float)u - t->rc_cx) / t->rc_fx;
        const float dy = ((float)v - t->rc_cy) / t->rc_fy;
        int prev_valid = 0, have_key = 0;
        float prev_sdf = 0.0f, prev_t = 0.0f, hit = 0.0f;
        int32_t ckey[3] = {0, 0, 0};
        int64_t cidx = -1;
        for (int i = 0; i < nsteps; ++i) {
            const float tt = t_min + (float)i * dt;
            const float pc[4] = { dx * tt, dy * tt, tt, 1.0f };
            float pw[4];
            vho_mat4_mul_vec4(pose, pc, pw);
            int32_t vox[3], key[3];
            vho_world2voxel(pw, t->p.voxelSize, vox);
            vho_voxel2block(vox, t->p.voxelBlockSize, key);
            if (!have_key || key[0] != ckey[0] || key[1] != ckey[1] || key[2] != ckey[2]) {
                ckey[0] = key[0]; ckey[1] = key[1]; ckey[2] = key[2];
                cidx = lookup_block(t, key);
                have_key = 1;
            }
            if (cidx < 0) { prev_valid = 0; continue; }
            const int32_t lx = wrap_sub(vox[0], wrap_mul(key[0], 8));
            const int32_t ly = wrap_sub(vox[1], wrap_mul(key[1], 8));
            const int32_t lz = wrap_sub(vox[2], wrap_mul(key[2], 8));
            const vho_voxel *vol = t->view_blocks ? t->view_blocks : t->blocks;
            const vho_voxel s = vol[(size_t)t->table[cidx].ptr + (size_t)(lz * 64 + ly * 8 + lx)];
            if (!(s.weight > 0.0f)) { prev_valid = 0; continue; }
            if (prev_valid && prev_sdf > 0.0f && s.sdf <= 0.0f) {
                hit = prev_t + (dt * prev_sdf) / (prev_sdf - s.sdf);
                break;
            }
            prev_valid = 1; prev_sdf = s.sdf; prev_t = tt;
        }
        depth_out[(size_t)v * W + u] = hit;
    }
}

/* ------------------------------------------------------------------ */
/* raycast as a voxel DDA: the traversal the reference's shader intends  */
/* (raycastSDF.frag:121-177, Amanatides-Woo between the ray's two ends;  */
/* disabled and self-declared broken there, so the spec is this build's) */
/* ------------------------------------------------------------------ */
/* Ray of pixel (u,v): world point at camera depth t is o + D*t with o = pose translation and
 * D = R*((u-cx)/fx, (v-cy)/fy, 1).  In voxel-grid coordinates g(t) = G + E*t, G = o/voxelSize + 0.5,
 * E = D/voxelSize, and voxel i covers [i, i+1) of g, i.e. world [(i-0.5), (i+0.5))*voxelSize around
 * its centre i*voxelSize (what world2Voxel's rounding maps to i).  The ray visits, in order, every voxel it
 * passes through between t_min and t_max:
 *   - a crossing EVENT of axis a out of coordinate c happens at tnext_a(c) = (c - Gs_a) * (1/E_a), Gs_a = G_a - 1
 *     (E_a > 0: the plane is c + 1) or G_a (E_a < 0): a pure function of the integer coordinate, never accumulated (the shader's
 *     tMax += tDelta, :159-169, drifts), so the traversal is the merge of three monotone event sequences;
 *   - events are merged by (t, axis priority y < z < x): exactly the shader's choice at :156-170 (x only when
 *     strictly first, z before x on a tie, y before both);
 *   - an axis with |E_a| <= 1e-20 never steps (:141-148).
 * Every visited voxel is a sample with the voxel's own {sdf, weight} (un-interpolated, :101-105), placed at the
 * camera depth of the voxel's CENTRE (row 2 of the cofactor inverse of the pose, scaled to voxel units: a TSDF
 * value is what integrateDepthMap measured at that centre along the camera axis, VoxelUtils.cu:797-813; against
 * the analytic room this halves the depth error of a mid-segment placement); valid = block allocated and
 * weight > 0.  Surface = first pair of consecutive
 * valid samples with sdf_prev > 0 >= sdf_cur; depth = t_prev + (t_cur - t_prev)*sdf_prev / (sdf_prev - sdf_cur);
 * 0 = miss.  Optional normal of a hit: gradient of the TSDF at the second voxel of the pair (central
 * differences where both neighbours are valid, one-sided otherwise), normalised, in the CAMERA frame, w = 0
 * (the convention of calculateNormals, CameraTrackingUtils.cu:75-113: towards the camera); zeros when an axis
 * has no valid neighbour or the gradient vanishes.
 *
 * jumps != 0: an absent block is left in one go.  The state after the jump is computed from the merge order
 * itself -- the exit event is the first of the three block-boundary events, every other axis advances past
 * exactly those of its events that precede it -- so the visited voxel sequence outside the empty block and
 * hence the image are IDENTICAL to the plain walk (tests/test_raycast_dda_cpu.py asserts the
 * bits).  That is what allows the HIP kernel to skip absent blocks and empty 4x4x4-block macro cells through
 * hashed bitmaps whose stale or colliding bits only make it skip less. */
typedef struct {
    float G[3], E[3], invE[3];
    float Gs[3];                 /* G - 1 for an axis that steps up (its crossing plane is c + 1), G otherwise */
    int   s[3], active[3];
} dda_ray;

static float dda_tnext(const dda_ray *r, int a, int32_t c)
{
    if (!r->active[a]) return VHO_INF_F;
    return ((float)c - r->Gs[a]) * r->invE[a];
}

/* event (tb, axis b) is merged before event (ta, axis a), a != b */
static int dda_before(float tb, int b, float ta, int a)
{
    static const int prio[3] = { 2, 0, 1 };      /* x, y, z: raycastSDF.frag:156-170 */
    return tb < ta || (tb == ta && prio[b] < prio[a]);
}

static int dda_first(const float t[3])           /* raycastSDF.frag:156,161,166 */
{
    if (t[0] < t[1] && t[0] < t[2]) return 0;
    if (t[2] < t[1]) return 2;
    return 1;
}

static int32_t floor_div8(int32_t v) { return v >> 3; }      /* = voxel2Block for two's complement ints (:37-42) */

/* the voxel (block allocated, weight > 0)? */
static int dda_voxel(const vho_table *t, const int32_t v[3], vho_voxel *out)
{
    const int32_t key[3] = { floor_div8(v[0]), floor_div8(v[1]), floor_div8(v[2]) };
    const int64_t idx = lookup_block(t, key);
    if (idx < 0) return 0;
    const vho_voxel *vol = t->view_blocks ? t->view_blocks : t->blocks;
    *out = vol[(size_t)t->table[idx].ptr + (size_t)((v[2] & 7) * 64 + (v[1] & 7) * 8 + (v[0] & 7))];
    return out->weight > 0.0f;
}

#define VHO_DDA_MAX_STEPS (1L << 22)

void vho_raycast_dda(vho_table *t, const float pose[16], float t_min, float t_max, int jumps, float *depth_out,
                     float *normal_out /* W*H*4 or NULL */)
{
    const int W = t->width, H = t->height;
    const float vs = t->p.voxelSize;
    float inv[16];
    vho_invert4x4(pose, inv);                      /* cofactor inverse, as SDF_Hashtable::integrate takes it */
    const float zrow[4] = { inv[8] * vs, inv[9] * vs, inv[10] * vs, inv[11] };
#pragma omp parallel for schedule(dynamic, 4)
    for (int v = 0; v < H; ++v)
    for (int u = 0; u < W; ++u) {
        const float dx = ((float)u - t->rc_cx) / t->rc_fx;
        const float dy = ((float)v - t->rc_cy) / t->rc_fy;
        dda_ray r;
        int32_t c[3];
        float tn[3];
        for (int a = 0; a < 3; ++a) {
            const float D = pose[4*a+0] * dx + pose[4*a+1] * dy + pose[4*a+2];
            r.G[a] = pose[4*a+3] / vs + 0.5f;
            r.E[a] = D / vs;
            r.active[a] = fabsf(r.E[a]) > 1.0e-20f;
            r.invE[a] = r.active[a] ? 1.0f / r.E[a] : 0.0f;
            r.s[a] = r.E[a] > 0.0f ? 1 : -1;
            r.Gs[a] = r.E[a] > 0.0f ? r.G[a] - 1.0f : r.G[a];
            c[a] = vho_float2int_rz(floorf(r.G[a] + r.E[a] * t_min));
        }
        for (int a = 0; a < 3; ++a) tn[a] = dda_tnext(&r, a, c[a]);
        float hit = 0.0f, prev_sdf = 0.0f, prev_t = 0.0f;
        int prev_valid = 0, have_key = 0, found = 0;
        int32_t ckey[3] = {0, 0, 0};
        int64_t cidx = -1;
        for (long it = 0; it < VHO_DDA_MAX_STEPS; ++it) {
            const int32_t key[3] = { floor_div8(c[0]), floor_div8(c[1]), floor_div8(c[2]) };
            if (!have_key || key[0] != ckey[0] || key[1] != ckey[1] || key[2] != ckey[2]) {
                ckey[0] = key[0]; ckey[1] = key[1]; ckey[2] = key[2];
                cidx = lookup_block(t, key);
                have_key = 1;
            }
            if (cidx < 0 && jumps) {
                int32_t cs[3];
                float te[3];
                for (int a = 0; a < 3; ++a) {
                    cs[a] = wrap_add(wrap_mul(key[a], 8), r.s[a] > 0 ? 7 : 0);     /* last coordinate inside the block */
                    te[a] = dda_tnext(&r, a, cs[a]);
                }
                const int x = dda_first(te);
                if (!(te[x] < t_max)) break;                   /* the ray ends inside the empty block */
                for (int b = 0; b < 3; ++b) {
                    if (b == x) continue;
                    while (c[b] != cs[b] && dda_before(dda_tnext(&r, b, c[b]), b, te[x], x)) c[b] = wrap_add(c[b], r.s[b]);
                }
                c[x] = wrap_add(cs[x], r.s[x]);
                for (int a = 0; a < 3; ++a) tn[a] = dda_tnext(&r, a, c[a]);
                prev_valid = 0;
                continue;
            }
            const int a = dda_first(tn);
            const float t_out = tn[a];
            if (cidx >= 0) {
                const vho_voxel *vol = t->view_blocks ? t->view_blocks : t->blocks;
                const vho_voxel sv = vol[(size_t)t->table[cidx].ptr + (size_t)((c[2] & 7) * 64 + (c[1] & 7) * 8 + (c[0] & 7))];
                if (sv.weight > 0.0f) {
                    /* the sample sits at the voxel's centre: its camera depth (row 2 of the inverse pose) */
                    const float t_mid = ((zrow[0] * (float)c[0] + zrow[1] * (float)c[1]) + zrow[2] * (float)c[2]) + zrow[3];
                    if (prev_valid && prev_sdf > 0.0f && sv.sdf <= 0.0f) {
                        hit = prev_t + ((t_mid - prev_t) * prev_sdf) / (prev_sdf - sv.sdf);
                        found = 1;
                        break;
                    }
                    prev_valid = 1; prev_sdf = sv.sdf; prev_t = t_mid;
                } else prev_valid = 0;
            } else prev_valid = 0;
            if (!(t_out < t_max)) break;
            c[a] = wrap_add(c[a], r.s[a]);
            tn[a] = dda_tnext(&r, a, c[a]);
        }
        depth_out[(size_t)v * W + u] = hit;
        if (!normal_out) continue;
        float n[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
        if (found) {
            vho_voxel here, sp, sm;
            float g[3];
            int ok = dda_voxel(t, c, &here);           /* (the hit voxel itself: valid by construction) */
            for (int a = 0; a < 3 && ok; ++a) {
                int32_t vp[3] = { c[0], c[1], c[2] }, vm[3] = { c[0], c[1], c[2] };
                vp[a] = wrap_add(vp[a], 1);
                vm[a] = wrap_sub(vm[a], 1);
                const int hp = dda_voxel(t, vp, &sp), hm = dda_voxel(t, vm, &sm);
                if (hp && hm) g[a] = (sp.sdf - sm.sdf) * 0.5f;
                else if (hp) g[a] = sp.sdf - here.sdf;
                else if (hm) g[a] = here.sdf - sm.sdf;
                else ok = 0;
            }
            if (ok) {
                const float len = sqrtf(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
                if (len > 0.0f) {
                    const float w[3] = { g[0] / len, g[1] / len, g[2] / len };
                    for (int i = 0; i < 3; ++i)          /* R^T * w: world -> camera */
                        n[i] = pose[0*4+i] * w[0] + pose[1*4+i] * w[1] + pose[2*4+i] * w[2];
                }
            }
        }
        memcpy(normal_out + 4 * ((size_t)v * W + u), n, sizeof n);
    }
}

/* ------------------------------------------------------------------ */
/* block silhouettes: the reference's one       */
/* working render pass, SDFRenderer::drawToFrontAndBack                 */
/* (SDFRenderer.cpp:165-208, depthWrite.*; back layer: notes.md:3-16)   */
/* ------------------------------------------------------------------ */
static int ray_box(const float o[3], const float d[3], const float lo[3], const float hi[3], float *t_near, float *t_far)
{
    float tn = -3.0e38f, tf = 3.0e38f;
    for (int a = 0; a < 3; ++a) {
        if (d[a] == 0.0f) {
            if (o[a] < lo[a] || o[a] > hi[a]) return 0;
            continue;
        }
        const float t0 = (lo[a] - o[a]) / d[a], t1 = (hi[a] - o[a]) / d[a];
        tn = fmaxf(tn, fminf(t0, t1));
        tf = fminf(tf, fmaxf(t0, t1));
    }
    *t_near = tn;
    *t_far = tf;
    return tn <= tf;
}

/* Per pixel the camera depth at which its ray (origin = pose translation, direction per unit
 * camera depth = R*(dx,dy,1) with the raycast intrinsics) enters the nearest and leaves the farthest
 * cube of any allocated block -- block k is the world box [8k, 8k+8]*voxelSize, the unit cube the
 * reference scales by 0.16 (Application.cpp:130-132) -- clipped to [t_min, t_max]; 0 = none. */
void vho_render_blocks(const vho_table *t, const float pose[16], float t_min, float t_max, float *front, float *back)
{
    const int W = t->width, H = t->height;
    float inv[16];
    vho_invert4x4(pose, inv);
    for (size_t i = 0; i < (size_t)W * H; ++i) { front[i] = INFINITY; back[i] = 0.0f; }
    const float o[3] = { pose[3], pose[7], pose[11] };
    const size_t n = (size_t)(t->bucket_hi - t->bucket_lo) * t->p.bucketSize;
    for (size_t e = 0; e < n; ++e) {
        if (t->table[e].ptr == VHO_FREE_BLOCK) continue;
        float lo[3], hi[3];
        for (int a = 0; a < 3; ++a) {
            lo[a] = (float)wrap_mul(t->table[e].pos[a], 8) * t->p.voxelSize;
            hi[a] = ((float)wrap_mul(t->table[e].pos[a], 8) + 8.0f) * t->p.voxelSize;
        }
        /* a generous screen bounding box keeps this loop short; any superset of the hit pixels gives the same image */
        float zmin = 3.0e38f, zmax = -3.0e38f, umin = 3.0e38f, umax = -3.0e38f, vmin = 3.0e38f, vmax = -3.0e38f;
        for (int c = 0; c < 8; ++c) {
            const float w[4] = { (c & 1) ? hi[0] : lo[0], (c & 2) ? hi[1] : lo[1], (c & 4) ? hi[2] : lo[2], 1.0f };
            float p[4];
            vho_mat4_mul_vec4(inv, w, p);
            if (p[2] < zmin) zmin = p[2];
            if (p[2] > zmax) zmax = p[2];
            const float z = p[2] > 1.0e-6f ? p[2] : 1.0e-6f;
            const float u = t->rc_fx * p[0] / z + t->rc_cx, v = t->rc_fy * p[1] / z + t->rc_cy;
            if (u < umin) umin = u;
            if (u > umax) umax = u;
            if (v < vmin) vmin = v;
            if (v > vmax) vmax = v;
        }
        if (zmax < t_min - 2.0f * t->p.voxelSize || zmin > t_max + 2.0f * t->p.voxelSize) continue;   /* out of the depth range */
        int x0 = 0, x1 = W - 1, y0 = 0, y1 = H - 1;
        if (zmin > 0.1f) {
            if (umax < -8.0f || vmax < -8.0f || umin > (float)W + 8.0f || vmin > (float)H + 8.0f) continue;
            x0 = (int)floorf(umin) - 6; if (x0 < 0) x0 = 0;
            y0 = (int)floorf(vmin) - 6; if (y0 < 0) y0 = 0;
            x1 = (int)ceilf(fminf(umax, 1.0e6f)) + 6; if (x1 > W - 1) x1 = W - 1;
            y1 = (int)ceilf(fminf(vmax, 1.0e6f)) + 6; if (y1 > H - 1) y1 = H - 1;
        }
        for (int py = y0; py <= y1; ++py)
        for (int px = x0; px <= x1; ++px) {
            const float dx = ((float)px - t->rc_cx) / t->rc_fx, dy = ((float)py - t->rc_cy) / t->rc_fy;
            const float d[3] = { pose[0] * dx + pose[1] * dy + pose[2], pose[4] * dx + pose[5] * dy + pose[6],
                                 pose[8] * dx + pose[9] * dy + pose[10] };
            float tn, tf;
            if (!ray_box(o, d, lo, hi, &tn, &tf)) continue;
            if (tf < t_min || tn > t_max) continue;
            /* (+ 0.0f: a face through the camera centre gives t = -0; the images hold +0 for it, so that
             * which of two equal zeros a pixel keeps does not depend on the order of the blocks) */
            const float f = fmaxf(tn, t_min) + 0.0f, k = fminf(tf, t_max) + 0.0f;
            float *pf = front + (size_t)py * W + px, *pb = back + (size_t)py * W + px;
            if (f < *pf) *pf = f;
            if (k > *pb) *pb = k;
        }
    }
    for (size_t i = 0; i < (size_t)W * H; ++i) if (front[i] == INFINITY) front[i] = 0.0f;
}

/* ------------------------------------------------------------------ */
/* block deletion / garbage collection (build extension, SURVEY.md      */
/* 8(f) next #4; DESIGN.md "deletion")                                  */
/* ------------------------------------------------------------------ */
/* The reference's deleteVoxelEntry (VoxelUtils.cu:544-604) is unreachable and frees the
 * block of the first FREE slot it meets; its heap push is removeSingleBlockInHeap
 * (:336-341).  Built "done correctly" as the paper the demo follows does it (Niessner et
 * al. 2013, section 4.4 "garbage collection"):
 *   - delete(key): find the entry of `key` in its bucket; zero its 512 voxels (blocks are
 *     handed out zeroed); push ptr/512 on the heap (:339-340); remove the
 *     entry and close the gap by moving the later entries of the bucket down in order, so
 *     a bucket's entries stay a prefix of its slots (insertVoxelEntry :421-456 stops at the
 *     first free slot; with a hole before an entry it would insert a duplicate);
 *   - collect(threshold): every entry of the compact list (the blocks the last flatten saw)
 *     whose voxels have max weight == 0, or min |sdf| over the voxels with weight > 0
 *     >= threshold, is deleted.  The compact list is empty afterwards. */

static void free_entry_at(vho_table *t, uint32_t local_bucket, uint32_t slot)
{
    const uint32_t bs = t->p.bucketSize;
    vho_entry *bucket = t->table + (size_t)local_bucket * bs;
    const int32_t ptr = bucket[slot].ptr;
    memset(t->blocks + ptr, 0, 512 * sizeof(vho_voxel));
    t->heap[++t->heap_counter] = (uint32_t)(ptr / 512);              /* :338-340 */
    uint32_t s = slot;
    for (; s + 1 < bs && bucket[s + 1].ptr != VHO_FREE_BLOCK; ++s) bucket[s] = bucket[s + 1];
    reset_entries(bucket + s, 1);
}

/* Deletion with the overflow list on (the reference's tail, VoxelUtils.cu:578-602, needs the
 * missing beforeThis(); completed as in the paper): no compaction -- a freed slot simply becomes
 * free -- except that a bucket's last slot, which heads its chain, is refilled with the first
 * chained entry when it is deleted while the chain is not empty (so "last slot free" always means
 * "no chain"); a chained entry is unlinked from its predecessor (prev.offset = curr.offset, :594).
 * The outcome does not depend on the order in which a set of keys is deleted. */
static void release_block(vho_table *t, int32_t ptr)
{
    memset(t->blocks + ptr, 0, 512 * sizeof(vho_voxel));
    t->heap[++t->heap_counter] = (uint32_t)(ptr / 512);              /* :338-340 */
}

static int delete_entry_overflow(vho_table *t, const int32_t key[3])
{
    const uint32_t bs = t->p.bucketSize;
    const uint32_t hg = vho_hash(key[0], key[1], key[2], t->p.numBuckets);
    if (hg < t->bucket_lo || hg >= t->bucket_hi) return 0;
    const uint32_t h = hg - t->bucket_lo;
    const size_t last = (size_t)h * bs + bs - 1;
    int64_t prev;
    const int64_t at = find_overflow(t, key, h, &prev);
    if (at < 0) return 0;
    vho_entry *e = &t->table[at];
    release_block(t, e->ptr);
    if ((size_t)at == last && e->offset != 0) {                     /* chain head with followers: pull the first one in */
        size_t base, n;
        chain_segment(t, h, &base, &n);
        vho_entry *nx = &t->table[chain_slot(last, e->offset, base, n)];
        *e = *nx;                                                    /* pos, ptr and its link to the rest of the chain */
        reset_entries(nx, 1);
    } else if (prev >= 0) {                                          /* chained entry */
        t->table[prev].offset = e->offset;                           /* :594 */
        reset_entries(e, 1);
    } else {
        reset_entries(e, 1);                                         /* a slot of the home bucket (a last slot here has offset 0) */
    }
    return 1;
}

/* keys: n x {x,y,z,ignored}.  Returns the number of blocks freed (absent keys are skipped). */
int vho_delete_blocks(vho_table *t, const int32_t *keys, int n)
{
    const uint32_t bs = t->p.bucketSize;
    int freed = 0;
    if (t->overflow) {
        for (int i = 0; i < n; ++i) freed += delete_entry_overflow(t, keys + 4 * (size_t)i);
        t->compact_counter = 0;
        return freed;
    }
    for (int i = 0; i < n; ++i) {
        const int32_t *k = keys + 4 * (size_t)i;
        const uint32_t hg = vho_hash(k[0], k[1], k[2], t->p.numBuckets);
        if (hg < t->bucket_lo || hg >= t->bucket_hi) continue;
        const uint32_t h = hg - t->bucket_lo;
        for (uint32_t sl = 0; sl < bs; ++sl) {
            const vho_entry *e = &t->table[(size_t)h * bs + sl];
            if (e->ptr == VHO_FREE_BLOCK) break;
            if (e->pos[0] == k[0] && e->pos[1] == k[1] && e->pos[2] == k[2]) {
                free_entry_at(t, h, sl);
                ++freed;
                break;
            }
        }
    }
    t->compact_counter = 0;
    return freed;
}

int vho_garbage_collect(vho_table *t, float sdf_threshold)
{
    const int n = t->compact_counter;
    int32_t *keys = (int32_t *)malloc((size_t)(n > 0 ? n : 1) * 4 * sizeof(int32_t));
    int m = 0;
    for (int b = 0; b < n; ++b) {
        const vho_entry *e = &t->compact[b];
        float min_abs = VHO_INF_F, max_w = 0.0f;
        for (int v = 0; v < 512; ++v) {
            const vho_voxel x = t->blocks[e->ptr + v];
            if (x.weight > 0.0f) min_abs = fminf(min_abs, fabsf(x.sdf));
            max_w = fmaxf(max_w, x.weight);
        }
        if (max_w == 0.0f || min_abs >= sdf_threshold) {
            keys[4 * m + 0] = e->pos[0]; keys[4 * m + 1] = e->pos[1]; keys[4 * m + 2] = e->pos[2]; keys[4 * m + 3] = 0;
            ++m;
        }
    }
    const int freed = vho_delete_blocks(t, keys, m);
    free(keys);
    return freed;
}

/* ------------------------------------------------------------------ */
/* raycast over shards: replicate the blocks a view can touch           */
/* (build extension, DESIGN.md section 6 "raycast")     */
/* ------------------------------------------------------------------ */

/* Conservative "can a ray of this view sample the block" test.  A ray sample lies in
 * the pyramid {t_min <= z <= t_max, pixel 0..W-1 x 0..H-1} of the view; the voxels of
 * block k span a cube of half-diagonal 4*sqrt(3) = 6.93 voxels around its centre
 * (8k+3.5)*voxelSize.  The block is kept when its centre is within 7 voxels of every
 * bounding plane of that pyramid: a superset of the blocks the rays touch, so a table
 * that holds exactly these blocks raycasts like the whole table.  The view's frustum
 * constants are prepared once on the host in this order (HIP side: vh_api.hip
 * make_view_frustum, same operations, -ffp-contract=off). */
void vho_view_frustum(const vho_table *t, const float pose[16], float t_min, float t_max, float f[22])
{
    float inv[16];
    vho_invert4x4(pose, inv);
    memcpy(f, inv, 12 * sizeof(float));
    const float r = 7.0f * t->p.voxelSize;
    const float a0 = (0.0f - t->rc_cx) / t->rc_fx, a1 = ((float)(t->width - 1) - t->rc_cx) / t->rc_fx;
    const float b0 = (0.0f - t->rc_cy) / t->rc_fy, b1 = ((float)(t->height - 1) - t->rc_cy) / t->rc_fy;
    const float a[4] = { a0, a1, b0, b1 };
    for (int i = 0; i < 4; ++i) {
        f[12 + i] = a[i];
        f[16 + i] = -(r * sqrtf(1.0f + a[i] * a[i]));
    }
    f[20] = t_min - r;
    f[21] = t_max + r;
}

int vho_view_holds_block(const vho_table *t, const float f[22], const int32_t key[3])
{
    float c[4], pc[3];
    for (int a = 0; a < 3; ++a) c[a] = ((float)wrap_mul(key[a], 8) + 3.5f) * t->p.voxelSize;
    for (int r = 0; r < 3; ++r)
        pc[r] = f[4 * r + 0] * c[0] + f[4 * r + 1] * c[1] + f[4 * r + 2] * c[2] + f[4 * r + 3];
    if (!(pc[2] >= f[20] && pc[2] <= f[21])) return 0;
    if (!(pc[0] - f[12] * pc[2] >= f[16])) return 0;      /* left   */
    if (!(f[13] * pc[2] - pc[0] >= f[17])) return 0;      /* right  */
    if (!(pc[1] - f[14] * pc[2] >= f[18])) return 0;      /* top    */
    if (!(f[15] * pc[2] - pc[1] >= f[19])) return 0;      /* bottom */
    return 1;
}

/* Every allocated entry of this table (shard) the view can touch, as records
 * {pos[3], 0, 512 voxels} (4112 bytes each), table order.  Returns the number of such
 * entries; only the first `capacity` are written. */
int vho_export_view(const vho_table *t, const float pose[16], float t_min, float t_max,
                    uint8_t *records, int capacity)
{
    float f[22];
    vho_view_frustum(t, pose, t_min, t_max, f);
    const size_t n = (size_t)(t->bucket_hi - t->bucket_lo) * t->p.bucketSize;
    int count = 0;
    for (size_t i = 0; i < n; ++i) {
        const vho_entry *e = &t->table[i];
        if (e->ptr == VHO_FREE_BLOCK || !vho_view_holds_block(t, f, e->pos)) continue;
        if (count < capacity) {
            uint8_t *rec = records + (size_t)VHO_VIEW_RECORD_BYTES * count;
            const int32_t head[4] = { e->pos[0], e->pos[1], e->pos[2], 0 };
            memcpy(rec, head, 16);
            memcpy(rec + 16, t->blocks + e->ptr, 512 * sizeof(vho_voxel));
        }
        ++count;
    }
    return count;
}

/* Replace the contents of `view` (an unsharded table of the same numBuckets / bucketSize,
 * used for nothing else) by `count` records gathered from the shards.  The voxels stay in
 * `records`, which must outlive the raycasts.  Returns the number of records that found no
 * free slot in their bucket (0 when the records come from one logical table of the same
 * geometry: a bucket of the view then holds a subset of the same logical bucket). */
int vho_import_view(vho_table *view, const uint8_t *records, int count)
{
    const uint32_t bs = view->p.bucketSize;
    reset_entries(view->table, (size_t)view->p.numBuckets * bs);
    view->view_blocks = (const vho_voxel *)records;
    int dropped = 0;
    for (int i = 0; i < count; ++i) {
        int32_t head[4];
        memcpy(head, records + (size_t)VHO_VIEW_RECORD_BYTES * i, 16);
        const uint32_t h = vho_hash(head[0], head[1], head[2], view->p.numBuckets);
        uint32_t s = 0;
        while (s < bs && view->table[(size_t)h * bs + s].ptr != VHO_FREE_BLOCK) ++s;
        const int32_t ptr = i * (VHO_VIEW_RECORD_BYTES / 8) + 2;   /* in voxels from the start of records */
        if (s == bs) {
            /* with the overflow list on, the shards' table holds chains: the record goes behind its full
             * bucket like an insertion (insert_entry_overflow), without the locks */
            if (!view->overflow) { ++dropped; continue; }
            const uint32_t L = view->p.attachedLinkedListSize;
            const size_t last = (size_t)h * bs + bs - 1;
            size_t base, n, at = last;
            chain_segment(view, h, &base, &n);
            uint32_t links = 0;
            int ended = 0;
            for (uint32_t iter = 0; iter < L; ++iter) {
                if (view->table[at].offset == 0) { ended = 1; break; }
                at = chain_slot(last, view->table[at].offset, base, n);
                ++links;
            }
            int64_t target = -1;
            int32_t tj = 0;
            if (ended && L >= 2 && links + 1 <= L - 1)
                for (int32_t j = 1; j < 10; ++j) {
                    const size_t q = chain_slot(last, j, base, n);
                    if (q % bs == bs - 1) continue;
                    if (view->table[q].ptr == VHO_FREE_BLOCK) { target = (int64_t)q; tj = j; break; }
                }
            if (target < 0) { ++dropped; continue; }
            vho_entry *e = &view->table[target];
            e->pos[0] = head[0]; e->pos[1] = head[1]; e->pos[2] = head[2];
            e->ptr = ptr;
            e->offset = view->table[last].offset;
            view->table[last].offset = tj;
            continue;
        }
        vho_entry *e = &view->table[(size_t)h * bs + s];
        e->pos[0] = head[0]; e->pos[1] = head[1]; e->pos[2] = head[2];
        e->ptr = ptr;
        e->offset = 0;
    }
    return dropped;
}

/* ------------------------------------------------------------------ */
/* bucket-range sharding (build extension)         */
/* ------------------------------------------------------------------ */

/* Key generation half of allocBlocksKernel (VoxelUtils.cu:606-636,673) for the
 * camera whose pose is set: every valid pixel whose block passes the frustum
 * test yields {x,y,z,rank}, rank = camera_id<<24 | launch rank; runs of equal
 * keys along an image row collapse to their first pixel.  Records are binned
 * by owning shard: bin s = bins[s*capacity ...], record 0 is the header
 * {count,0,0,0}, records 1..count the keys.  Returns the largest count (a
 * count > capacity-1 means that bin overflowed). */
int vho_generate_keys(vho_table *t, const float *verts, uint32_t camera_id, int num_shards,
                      int32_t *bins, int capacity)
{
    const int W = t->width, H = t->height;
    const uint32_t per = (t->p.numBuckets + (uint32_t)num_shards - 1u) / (uint32_t)num_shards;
    int worst = 0;
    for (int s = 0; s < num_shards; ++s) memset(bins + (size_t)4 * s * capacity, 0, 4 * sizeof(int32_t));
    float step;
    const int nS = band_samples(t, &step);
    for (int y = 0; y < H; ++y) {
        int have_prev[VHO_MAX_BAND_SAMPLES];
        int32_t prev[VHO_MAX_BAND_SAMPLES][3];
        memset(have_prev, 0, sizeof have_prev);
        for (int x = 0; x < W; ++x) {
            const float *v = verts + 4 * ((size_t)y * W + x);
            int32_t pk[VHO_MAX_BAND_SAMPLES][3];
            uint8_t valid[VHO_MAX_BAND_SAMPLES];
            int n = 0;
            if (v[2] != 0.0f) n = pixel_keys(t, verts, x, y, nS, step, pk, valid);
            for (int k = 0; k < VHO_MAX_BAND_SAMPLES; ++k) {
                const int32_t *key = pk[k];
                const int want = k < n && valid[k] && vho_block_in_frustum(t, key);
                if (!want) { have_prev[k] = 0; continue; }
                if (have_prev[k] && key[0] == prev[k][0] && key[1] == prev[k][1] && key[2] == prev[k][2]) continue;
                have_prev[k] = 1; prev[k][0] = key[0]; prev[k][1] = key[1]; prev[k][2] = key[2];
                const uint32_t h = vho_hash(key[0], key[1], key[2], t->p.numBuckets);
                int32_t *bin = bins + (size_t)4 * (h / per) * capacity;
                const int slot = ++bin[0];
                if (slot > worst) worst = slot;
                if (slot < capacity) {
                    int32_t *r = bin + 4 * slot;
                    r[0] = key[0]; r[1] = key[1]; r[2] = key[2];
                    /* camera, then launch order, then sample index decide who wins a bucket */
                    r[3] = (int32_t)((camera_id << 27) | (vho_launch_rank(x, y, W) << 6) | (uint32_t)k);
                }
            }
        }
    }
    return worst;
}

typedef struct { int32_t k[4]; } key_rec;
static int cmp_rank(const void *a, const void *b)
{
    const uint32_t ra = (uint32_t)((const key_rec *)a)->k[3], rb = (uint32_t)((const key_rec *)b)->k[3];
    return (ra > rb) - (ra < rb);
}

/* Insert the keys of num_bins received bins under ONE lock epoch: contenders
 * are served in rank order, which is what the min-rank claim of the HIP path
 * resolves to.  Call vho_reset_mutexes first. */
int vho_insert_bins(vho_table *t, const int32_t *bins, int num_bins, int capacity)
{
    size_t total = 0;
    for (int b = 0; b < num_bins; ++b) {
        int n = bins[(size_t)4 * b * capacity];
        if (n > capacity - 1) n = capacity - 1;
        total += (size_t)n;
    }
    key_rec *all = (key_rec *)malloc(sizeof(key_rec) * (total ? total : 1));
    size_t m = 0;
    for (int b = 0; b < num_bins; ++b) {
        const int32_t *bin = bins + (size_t)4 * b * capacity;
        int n = bin[0];
        if (n > capacity - 1) n = capacity - 1;
        for (int i = 1; i <= n; ++i) memcpy(all[m++].k, bin + 4 * i, sizeof(key_rec));
    }
    qsort(all, total, sizeof(key_rec), cmp_rank);
    for (size_t i = 0; i < total; ++i) insert_entry(t, all[i].k);
    free(all);
    return (int)total;
}

/* Camera packet: 16 floats pose, 16 floats inverse pose, W*H camera-z plane. */
void vho_write_packet(vho_table *t, const float *verts, float *packet)
{
    memcpy(packet, t->p.global_transform, 16 * sizeof(float));
    memcpy(packet + 16, t->p.inv_global_transform, 16 * sizeof(float));
    const size_t n = (size_t)t->width * t->height;
    for (size_t i = 0; i < n; ++i) packet[32 + i] = verts[4 * i + 2];
}

/* For each camera packet in order: flatten against that camera's frustum and
 * run the TSDF update from its depth plane -- the flatten/integrate half of
 * SDF_Hashtable::integrate applied camera by camera to this shard. */
int vho_integrate_packets(vho_table *t, int num_cams, const float *packets)
{
    const size_t stride = 32 + (size_t)t->width * t->height;
    int total = 0;
    for (int c = 0; c < num_cams; ++c) {
        const float *pk = packets + stride * c;
        memcpy(t->p.global_transform, pk, 16 * sizeof(float));
        memcpy(t->p.inv_global_transform, pk + 16, 16 * sizeof(float));
        int occ = vho_flatten(t);
        if (occ > 0) integrate_depth(t, pk + 32, 1);
        total += occ;
    }
    return total;
}

uint32_t vho_bucket_lo(const vho_table *t) { return t->bucket_lo; }
uint32_t vho_bucket_hi(const vho_table *t) { return t->bucket_hi; }

/* ------------------------------------------------------------------ */
/* depth pre-processing: the step that makes   */
/* the vertex / normal maps integrate() consumes                        */
/* ------------------------------------------------------------------ */

/* calculateVertexPositions, CameraTrackingUtils.cu:50-73: depth = d / 5000.0f (TUM
 * convention), point = (K_inv * (x, y, 1)) * depth, w = 1; an invalid depth of 0
 * gives (0,0,0,1).  k_inv is the 3x3 the reference uploads with SetCameraIntrinsic
 * (:218-222), read row-major by float3x3::operator*. */
static void vertex_from_depth(const uint16_t *depth, const float k_inv[9], int W, int x, int y, float out[3])
{
    const float d = (float)depth[(size_t)y * W + x] / 5000.0f;
    const float c[3] = { (float)x, (float)y, 1.0f };
    float p[3];
    mat3_mul_vec3(k_inv, c, p);
    out[0] = p[0] * d; out[1] = p[1] * d; out[2] = p[2] * d;
}

/* preProcess, CameraTrackingUtils.cu:115-120 = calculateVertexPositions (:50-73) +
 * calculateNormals (:75-113): central differences, cross product, normalised with a
 * true divide; zero where a neighbour is missing (.x == 0) or on the image border. */
void vho_preprocess(const uint16_t *depth, const float k_inv[9], int W, int H,
                    float *positions /* W*H*4 */, float *normals /* W*H*4 */)
{
    for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
        float *v = positions + 4 * ((size_t)y * W + x);
        vertex_from_depth(depth, k_inv, W, x, y, v);
        v[3] = 1.0f;
    }
    for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
        float *n = normals + 4 * ((size_t)y * W + x);
        n[0] = n[1] = n[2] = n[3] = 0.0f;
        if (!(x > 0 && x < W - 1 && y > 0 && y < H - 1)) continue;                   /* :95 */
        const float *CC = positions + 4 * ((size_t)y * W + x);
        const float *PC = positions + 4 * ((size_t)(y + 1) * W + x);
        const float *CP = positions + 4 * ((size_t)y * W + x + 1);
        const float *MC = positions + 4 * ((size_t)(y - 1) * W + x);
        const float *CM = positions + 4 * ((size_t)y * W + x - 1);
        if (!(CC[0] != 0 && PC[0] != 0 && CP[0] != 0 && MC[0] != 0 && CM[0] != 0)) continue;   /* :102 */
        const float a[3] = { PC[0] - MC[0], PC[1] - MC[1], PC[2] - MC[2] };
        const float b[3] = { CP[0] - CM[0], CP[1] - CM[1], CP[2] - CM[2] };
        const float c[3] = { a[1]*b[2] - a[2]*b[1], a[2]*b[0] - a[0]*b[2], a[0]*b[1] - a[1]*b[0] };   /* cross */
        const float l = sqrtf(c[0]*c[0] + c[1]*c[1] + c[2]*c[2]);                       /* length */
        if (l > 0.0f) { n[0] = c[0] / l; n[1] = c[1] / l; n[2] = c[2] / l; n[3] = 0.0f; }   /* :108-110 */
    }
}

/* ------------------------------------------------------------------ */
/* accessors                                                           */
/* ------------------------------------------------------------------ */
const vho_params *vho_get_params(const vho_table *t) { return &t->p; }
vho_entry *vho_hash_table(vho_table *t) { return t->table; }
vho_entry *vho_compact_table(vho_table *t) { return t->compact; }
int vho_compact_count(const vho_table *t) { return t->compact_counter; }
const uint32_t *vho_heap(const vho_table *t) { return t->heap; }
vho_voxel *vho_sdf_blocks(vho_table *t) { return t->blocks; }
int vho_heap_counter(const vho_table *t) { return t->heap_counter; }
const vho_frame_stats *vho_last_stats(const vho_table *t) { return &t->stats; }
