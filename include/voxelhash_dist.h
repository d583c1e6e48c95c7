/*
 * voxelhash_dist.h -- C-ABI of the multi-GPU host of libvoxelhash_hip.so: one logical voxel-hash table cut into
 * bucket ranges, one process (or thread) per GPU, the per-frame exchange on RCCL over xGMI.
 *
 * The reference's host is C++ (SDF_Hashtable.cpp:11-40) and drives ONE GPU.  This is the same integrate() for R
 * cameras into one table sharded over R GPUs: rank r owns buckets [r*per, (r+1)*per) -- entries, heap
 * and voxel blocks of every key that hashes there -- and holds camera r.  A step is a batch of `batch` multi-camera
 * frames (voxelhash.h, "bucket-range sharding"):
 *
 *     generate   this rank's frames -> block keys binned by owner + camera packets        (vh_generate_keys_*_batch)
 *     exchange   ncclAllToAll of the key bins, ncclAllGather of the packets               (RCCL, its own stream)
 *     apply      on the owner: lock epoch, insert, walk + TSDF update for all cameras      (vh_apply_frames_batch)
 *
 * software-pipelined inside the library over three HIP streams and three buffer sets with events (while exchange n travels
 * into one set, the frames of exchange n-1 are applied from the second, one launch per multi-camera frame, and the last
 * frame of exchange n-2 -- whose commit + TSDF update ride in the first launch of n-1's frames -- still reads its packets in
 * the third), as the Python host of round 2 did it with two (voxelhashing_demo_amd/dist.py: ShardedPipeline) -- but one C
 * call per exchange instead of a dozen Python calls and four torch collectives.  The table sees its
 * operations in the order of the plain step sequence, so results do not depend on the pipelining.
 * On shards large enough for it to pay (option "fused_generation") the generate step has no launches of its own: it runs as a role
 * of the frame launches that apply the exchange fed two calls earlier, over four buffer sets -- the table stream then carries
 * generation and table work, the collectives of exchange n overlap the frame launches of exchange n-1.
 *
 * RCCL is bound at run time (dlopen of librccl.so.1, or whatever copy the process has loaded already -- torch's): the
 * library has no link-time dependency on it, and single-GPU users never load it.  No torch types anywhere.
 */
#ifndef VOXELHASH_DIST_H
#define VOXELHASH_DIST_H

#include "voxelhash.h"

#ifdef __cplusplus
extern "C" {
#endif

#define VH_DIST_ID_BYTES 128            /* = NCCL_UNIQUE_ID_BYTES */

typedef struct vh_dist vh_dist;

typedef struct vh_dist_config {
    vh_config table;          /* the LOGICAL table (params.numBuckets = all buckets; numVoxelBlocks = per rank), image size,
                                 semantics, device of this rank */
    int32_t rank, world;      /* this rank's index and the number of ranks = cameras = shards (<= VH_MAX_CAMERAS) */
    int32_t batch;            /* frames per camera and exchange (>= 1) */
    int32_t key_capacity;     /* records per (camera, owner) key bin -- one bin for the whole batch (VH_BIN_PER_BATCH);
                                 0 = max(8192, 1.5 * batch * W*H/16 / world) */
    int32_t packet_format;    /* VH_PACKET_U16: frames are uint16 sensor images (k_inv used), packets carry the image;
                                 VH_PACKET_F32: frames are float4 vertex maps, packets carry a float camera-z plane */
    float   k_inv[9];         /* row-major K^-1 of the cameras (VH_PACKET_U16) */
} vh_dist_config;

/* ncclGetUniqueId: one rank calls it, every rank passes the same bytes to vh_dist_create (exchanged out of band: MPI,
 * a file, torch.distributed over gloo ...).  world == 1 needs no exchange. */
int vh_dist_unique_id(char id[VH_DIST_ID_BYTES]);

/* Loads RCCL and resolves its symbols without creating anything and without any collective: lets every rank find out
 * locally whether vh_dist_create can work before the ranks agree to call it (a rank that fails inside a collective
 * leaves the others waiting). */
int vh_dist_probe(void);

/* The transports.  vh_dist reaches its peers through two stream-ordered collectives only (all-to-all, all-gather), and
 * vh_dist_create picks who carries them from the id:
 *   - an id of vh_dist_unique_id (or an adopted ncclComm_t): RCCL over xGMI, one process per GPU -- the default;
 *   - an id of vh_dist_loopback_id: the LOOP-BACK transport, which joins the `world` vh_dist instances of ONE process,
 *     usually all on one device: peer buffers are copied with hipMemcpyAsync on the calling rank's stream, ordered by
 *     events.  Every collective call (vh_dist_step_batch, vh_dist_raycast) must then be made by `world` host threads,
 *     one per rank, as it would be by `world` processes (a rank that does not arrive within 120 s -- environment
 *     VOXELHASH_LOOPBACK_TIMEOUT_S -- fails the call for all).  Everything but the bytes' way across is the code the RCCL ranks run -- buffer sets, events, deferred
 *     frames -- which is what the transport is for: the N > 1 exchange under test on a single-GPU box. */
int vh_dist_loopback_id(char id[VH_DIST_ID_BYTES]);
/* "rccl" | "loopback" */
const char *vh_dist_transport_name(vh_dist *d);
/* rank and size as the transport itself reports them (ncclCommUserRank / ncclCommCount; loop-back: ranks that have joined) */
int vh_dist_comm_info(vh_dist *d, int32_t *rank, int32_t *world);
/* The form of the key generation THIS rank's exchanges run in (decided per rank at the first exchange and behind every
 * vh_dist_flush, option "fused_generation"): 1 = a role of the frame launches, 0 = launches of its own on a second stream.
 * Ranks may differ (the size rule looks at the rank's own shard); every form issues the same collectives in the same call.
 * Negative: an error code. */
int vh_dist_generation_form(vh_dist *d);

/* Creates this rank's shard (vh_create_shard over its bucket range), the communicator (ncclCommInitRank with `id`; or
 * adopts `nccl_comm`, an ncclComm_t the caller owns, when it is not NULL), streams, events and the exchange buffers.
 * Collective: every rank calls it. */
int vh_dist_create(const vh_dist_config *cfg, const char id[VH_DIST_ID_BYTES], void *nccl_comm, vh_dist **out);
/* Synchronises this rank's device first.  Ranks of a loop-back group share the events of their collectives: destroy them
 * only after every rank of the group has returned from vh_dist_flush (NativeGroup.close does), as the ranks of an RCCL
 * group must not be destroyed inside a collective their peers are still in. */
int vh_dist_destroy(vh_dist *d);

/* this rank's shard: counters, download, snapshot, options (set before the first step) ... go through voxelhash.h */
vh_context *vh_dist_shard(vh_dist *d);

/* STREAM CONTRACT.  vh_dist works on three private non-blocking streams.  Without a user stream the caller owns the
 * ordering: the frames passed to vh_dist_step_batch must be complete before the call and must stay untouched until a
 * later vh_dist_flush (or until the next-but-two vh_dist_step_batch has returned and the device has caught up), and the
 * image of vh_dist_raycast is valid after vh_dist_flush / a device synchronisation.  With a user stream
 * (enable != 0; `stream` may be the null stream) the calls are ordered against it like ordinary stream work:
 * vh_dist_step_batch reads the frames behind everything queued on `stream` so far and makes `stream` wait until they have
 * been consumed (work queued on it afterwards may overwrite them); vh_dist_raycast writes d_depth_out / d_lost behind
 * `stream`'s queued work and makes `stream` wait for the image. */
int vh_dist_set_user_stream(vh_dist *d, void *stream, int32_t enable);

/* One exchange: `batch` frames of THIS rank's camera (poses: batch*16 host floats; d_frames: host array of `batch`
 * device pointers -- uint16 sensor images or float4 vertex maps by packet_format).  Enqueues the generation and the
 * collectives of this exchange and the application of an EARLIER one -- the previous one, or with option "fused_generation"
 * (the default) the one before it, in whose frame launches this exchange's generation rides; returns without waiting.
 * Collective. */
int vh_dist_step_batch(vh_dist *d, const float *poses, const void *const *d_frames);
/* applies the exchange(s) in flight -- one, or two with "fused_generation" -- and waits for the three streams */
int vh_dist_flush(vh_dist *d);

/* Raycast of this rank's view through the WHOLE sharded table (voxelhash.h: vh_export_views_fixed / vh_import_views):
 * all-gather of the R view poses, one walk of the shard for all views, ncclAllToAll of fixed record slots, import into a
 * private view table, vh_raycast -- one stream, no host synchronisation.  capacity: records per (shard, view) slot range.
 * d_lost (device int32, nullable): records this view's sources selected beyond the capacity.  Collective. */
int vh_dist_raycast(vh_dist *d, const float pose[16], float t_min, float t_max, int32_t capacity, float *d_depth_out,
                    int32_t *d_lost);

/* The same round with the slot capacity chosen by the library and no holes: renders, gathers every rank's lost count, and
 * repeats the round for all ranks with more room while any view lost records (the slots are one size everywhere, so the
 * ranks decide together -- the first capacity too: the largest any rank proposes, gathered before the first round).  Synchronises the host.  d_normals_out (nullable): camera-frame normals of the hits, as
 * vh_raycast_normals.  capacity_used (nullable): the capacity that rendered every view whole.  Collective. */
int vh_dist_raycast_auto(vh_dist *d, const float pose[16], float t_min, float t_max, float *d_depth_out,
                         vh_float4 *d_normals_out, int32_t *capacity_used);

/* host seconds spent inside vh_dist_step_batch since creation / the number of calls (diagnostics for bench.py) */
int vh_dist_host_stats(vh_dist *d, double *seconds, uint64_t *calls);

/* Options of the exchange:
 *   "force_collectives" 0 | 1   with ONE rank nothing is exchanged -- the frames are applied straight from the send buffers, no
 *                               collective, no copy (vh_dist_transport_name still names the transport that WOULD carry them).
 *                               1 makes a one-rank group run ncclAllToAll / ncclAllGather all the same: the call sequence of an
 *                               R-GPU node, exercised where only one GPU is at hand.  Set before the first exchange or behind
 *                               vh_dist_flush.
 *   "fused_generation" 0|1|2    The key generation of an exchange as a role of the frame launches that apply the exchange two
 *                               calls earlier -- no launches of its own, no second stream; with one rank and no caller stream a
 *                               steady-state vh_dist_step_batch is `batch` kernel launches and no event operation.  0: never;
 *                               2: wherever the shard's frame launch can carry it; 1 (the default): where it also pays -- a
 *                               shard of more than 60 MB of table, whose walk is long enough to hide the role (C2's 2^20
 *                               buckets on one rank: 49 k against 46 k frames/s; cut 4 or 8 ways: separate launches win).
 *                               Fused, an exchange is applied by the SECOND call after the one that fed it (else by the next
 *                               call); vh_dist_flush applies whatever is in flight either way.  The launch cannot carry the
 *                               role with float packets, an allocation band, the overflow list or the walk-free launch of
 *                               "flatten_variant" 4 (looked at in the first exchange and behind every vh_dist_flush), nor in
 *                               the first two calls behind a flush or with batch > 8.  Same results bit for bit.  Set before
 *                               the first exchange or behind vh_dist_flush.  Environment VOXELHASH_DIST_FUSED: the default.
 *   "phase_timing" 0 | 1        timing events around the three phases of every exchange (vh_dist_phase_times).  Off by default.
 *   "raycast_auto_start" n      this rank's proposal for the slot capacity of vh_dist_raycast_auto's first round (default 4096;
 *                               the ranks take the largest proposal, so they need not agree on it) */
int vh_dist_set_option(vh_dist *d, const char *name, int32_t value);

/* Sums over the exchanges completed since the last reset (option "phase_timing"), microseconds on the device's clock:
 *   generate_us        the key-generation launch(es) of an exchange (its stream may share the GPU with frame launches); with
 *                      "fused_generation" the span of the frame launches that carried it -- the same launches apply_us times
 *   collectives_us     from the moment the exchange's collectives may start (generation done, receive buffers free) to their
 *                      completion: ncclAllToAll of the key bins + ncclAllGather of the packets (world 1: ~0, nothing is sent)
 *   apply_us           the frame launches of the exchange on the owner, first to last
 *   first_to_last_us   from the start of the generation to the end of the frame launches: the latency of one exchange (three
 *                      of them overlap; four with "fused_generation")
 *   host_enqueue_us    host time inside vh_dist_step_batch for as many calls
 * A measured scaling curve is read against these: the period of an exchange is max(generate, collectives, apply) when the
 * three streams overlap as designed, their sum when they do not. */
typedef struct vh_dist_phases {
    double generate_us, collectives_us, apply_us, first_to_last_us, host_enqueue_us;
    uint64_t exchanges;
} vh_dist_phases;
int vh_dist_phase_times(vh_dist *d, vh_dist_phases *out, int32_t reset);

/* Start-up check of the transport: a known pattern through the two collectives an exchange uses (all-to-all of one 64 KB
 * slice per peer, all-gather of 64 KB), compared on the device; fails loudly (VH_ERR_HIP, the count of wrong words in
 * vh_last_error) before anything is fused or timed.  Flushes first; leaves the exchange buffers empty.  Collective. */
int vh_dist_self_check(vh_dist *d);

#ifdef __cplusplus
}
#endif
#endif /* VOXELHASH_DIST_H */
