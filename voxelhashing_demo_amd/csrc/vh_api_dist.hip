// vh_api_dist.hip -- C-ABI, the multi-GPU host (include/voxelhash_dist.h): bucket-range shards, the per-frame exchange
// on RCCL, pipelined over three HIP streams.  Included by vh_api.hip (same translation unit: shares fail(), VH_HIP,
// DeviceGuard and the context).
#include <dlfcn.h>

#include <chrono>
#include <deque>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>

#include <rccl/rccl.h>          // types and prototypes only: the library is bound at run time (rccl_load)

#include "../../include/voxelhash_dist.h"
#ifndef VH_DIST_FUSED_DEFAULT
#define VH_DIST_FUSED_DEFAULT 1      // option "fused_generation" of a new vh_dist
#endif

// ---------------------------------------------------------------------------
// RCCL, bound at run time
// ---------------------------------------------------------------------------
// A process that already holds a copy of RCCL (torch's bundled librccl.so) must keep using THAT copy -- two collective
// libraries in one process each build their own view of the devices -- so the symbols are looked up in the already
// loaded image first (RTLD_NOLOAD), then in librccl.so.1 by name, then in the ROCm install.
namespace {
struct Rccl {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) getUniqueId = nullptr;
    decltype(&ncclCommInitRank) commInitRank = nullptr;
    decltype(&ncclCommDestroy) commDestroy = nullptr;
    decltype(&ncclAllToAll) allToAll = nullptr;
    decltype(&ncclAllGather) allGather = nullptr;
    decltype(&ncclGetErrorString) getErrorString = nullptr;
    decltype(&ncclCommCount) commCount = nullptr;
    decltype(&ncclCommUserRank) commUserRank = nullptr;
    bool ok = false;
};
Rccl g_rccl;
std::mutex g_rccl_mutex;          // vh_dist_create may be called by one thread per GPU

int rccl_load()
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.ok) return VH_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *lib = nullptr;
    for (const char *n : names)
        if ((lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD)) != nullptr) break;
    for (size_t i = 0; !lib && i < sizeof names / sizeof names[0]; ++i) lib = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!lib) return fail(VH_ERR_HIP, "librccl.so.1 could not be loaded (the multi-GPU host needs RCCL)");
    g_rccl.lib = lib;
#define VH_RCCL_SYM(field, name)                                                     \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(lib, name));       \
    if (!g_rccl.field) return fail(VH_ERR_HIP, "RCCL symbol missing: " name)
    VH_RCCL_SYM(getUniqueId, "ncclGetUniqueId");
    VH_RCCL_SYM(commInitRank, "ncclCommInitRank");
    VH_RCCL_SYM(commDestroy, "ncclCommDestroy");
    VH_RCCL_SYM(allToAll, "ncclAllToAll");
    VH_RCCL_SYM(allGather, "ncclAllGather");
    VH_RCCL_SYM(getErrorString, "ncclGetErrorString");
    VH_RCCL_SYM(commCount, "ncclCommCount");
    VH_RCCL_SYM(commUserRank, "ncclCommUserRank");
#undef VH_RCCL_SYM
    g_rccl.ok = true;
    return VH_OK;
}

int rccl_fail(const char *what, ncclResult_t r)
{
    g_last_error = what;
    g_last_error += ": ";
    g_last_error += g_rccl.getErrorString ? g_rccl.getErrorString(r) : "RCCL error";
    return VH_ERR_HIP;
}
#define VH_RCCL(call)                                         \
    do {                                                      \
        const ncclResult_t r_ = (call);                       \
        if (r_ != ncclSuccess) return rccl_fail(#call, r_);   \
    } while (0)
}  // namespace

// ---------------------------------------------------------------------------
// Transports: the two stream-ordered collectives the exchange is made of
// ---------------------------------------------------------------------------
// vh_dist talks to its peers through this table only.  Both calls are collective, enqueue on `s` and return without
// waiting for the device; when the work on `s` completes, `recv` holds every peer's part and `send` may be overwritten.
//   all_to_all:  recv[p] (bytes each) = peer p's send[rank]
//   all_gather:  recv[p] (bytes each) = peer p's send
// RCCL over xGMI is the default entry (one process per GPU).  The loop-back entry joins the R vh_dist instances of ONE
// process on ONE device -- one host thread per rank, like one process per GPU -- so that the code of vh_dist_step_batch /
// vh_dist_raycast runs with R > 1 on a single-GPU box: the three buffer sets, the generated / ready / first events, the
// deferred frame of pipeline_shards 2 and the fixed-slot view round are exercised exactly as over RCCL; only the bytes
// travel by hipMemcpyAsync instead of by a collective kernel.
struct Transport {
    const char *name;
    int (*all_to_all)(vh_dist *d, const void *send, void *recv, size_t bytes, hipStream_t s);
    int (*all_gather)(vh_dist *d, const void *send, void *recv, size_t bytes, hipStream_t s);
    void (*leave)(vh_dist *d);
};

namespace {
// One collective of a loop-back group: every rank publishes its buffers and an event behind what its stream has queued
// so far (`posted`: the send buffer is final), the ranks meet on the host, every rank queues its R copies behind the
// peers' `posted` events and records `done`; a second meeting, and every rank's stream waits for all `done` events --
// its send buffer has been read by everyone, which is what the completion of an RCCL collective means.
struct LoopGroup {
    std::mutex m;
    std::condition_variable cv;
    int world = 0, joined = 0, arrived = 0;
    uint64_t phase = 0, serial = 0;
    bool broken = false;
    struct Post {
        const void *send = nullptr;
        void *recv = nullptr;
        size_t bytes = 0;
        int kind = 0;
        hipEvent_t posted = nullptr, done = nullptr;
        bool present = false;
    } post[VH_MAX_CAMERAS];
};
std::mutex g_loop_mutex;
std::map<uint64_t, std::shared_ptr<LoopGroup>> g_loops;
uint64_t g_loop_serial = 0;
const char kLoopMagic[8] = {'V', 'H', 'L', 'O', 'O', 'P', 'B', 'K'};

bool loop_meet(LoopGroup *g)
{
    std::unique_lock<std::mutex> lk(g->m);
    if (g->broken) return false;
    const uint64_t ph = g->phase;
    if (++g->arrived == g->world) {
        g->arrived = 0;
        g->phase += 1;
        g->cv.notify_all();
        return true;
    }
    // a rank that never arrives (an error on its side, a host that drives the ranks from one thread) must not hang the rest
    const char *env = std::getenv("VOXELHASH_LOOPBACK_TIMEOUT_S");
    const int seconds = env && std::atoi(env) > 0 ? std::atoi(env) : 120;
    if (!g->cv.wait_for(lk, std::chrono::seconds(seconds), [&] { return g->phase != ph || g->broken; })) {
        g->broken = true;
        g->cv.notify_all();
        return false;
    }
    return !g->broken;
}
}  // namespace

// the view pose of a raycast round into device memory without a staging buffer: 16 floats in the kernel arguments
struct Pose16 { float m[16]; };
__global__ void dist_store_pose_kernel(const Pose16 p, float *__restrict__ out)
{
    if (threadIdx.x < 16) out[threadIdx.x] = p.m[threadIdx.x];
}
// records this view's sources selected beyond the slot capacity (the count travels in the first record's spare word)
__global__ void dist_lost_kernel(const uint8_t *__restrict__ records, int32_t numSources, int32_t capacity, int32_t *lost)
{
    if (threadIdx.x == 0) {
        int32_t n = 0;
        for (int s = 0; s < numSources; ++s) {
            const int32_t c = *reinterpret_cast<const int32_t *>(records + (size_t)s * capacity * sizeof(vh_view_record) + 12);
            n += c > capacity ? c - capacity : 0;
        }
        *lost = n;
    }
}

struct vh_dist {
    vh_dist_config cfg;
    vh_context *shard = nullptr;
    vh_context *view = nullptr;            // raycast over the shards: the private view table (first vh_dist_raycast)
    const Transport *transport = nullptr;  // chosen in vh_dist_create: RCCL unless the id names a loop-back group
    ncclComm_t comm = nullptr;
    bool ownComm = false;
    std::shared_ptr<LoopGroup> loop;       // loop-back transport: the group this rank has joined
    hipStream_t userStream = nullptr;      // vh_dist_set_user_stream: the caller's stream the frames / images are ordered against
    bool haveUser = false;
    hipEvent_t userEvent = nullptr, outEvent = nullptr;
    int device = 0;
    int capacity = 0;                      // records per key bin
    size_t packetUnits = 0;                // 4-byte units of one camera packet
    hipStream_t sGen = nullptr, sComm = nullptr, sTable = nullptr;
    // Three buffer sets: while exchange n travels into one, the frames of exchange n-1 are applied from the second, and the
    // last frame of exchange n-2 -- whose commit + TSDF update ride in the first launch of n-1's frames (pipeline_shards 2) --
    // still reads its packets in the third.
    // (fused generation, below: FOUR sets -- the generation of exchange n rides in the frame launches of exchange n-2, its collectives
    // travel beside the launches of n-1, it is applied with n+2's generation aboard, and its last frame's deferred half still reads
    // its packets in the first launch after that.  The separate-generation path keeps rotating over the first three.)
    static constexpr int kSets = 4, kSetsSeparate = 3;
    hipEvent_t generated[kSets] = {}, ready[kSets] = {}, first[kSets] = {};     // first: behind the first launch of the set's frames
    struct Set {
        int32_t *binsSend = nullptr, *binsRecv = nullptr;      // [world][batch][capacity][4]
        float *packet = nullptr, *packets = nullptr;          // [batch][P], [world][batch][P]
    } set[kSets];
    uint64_t count = 0;                    // exchanges fed
    int pending = -1;                      // buffer set whose exchange is in flight / landed but not applied
    // raycast round
    float *poseMine = nullptr, *poseAll = nullptr;
    vh_view_record *viewSend = nullptr, *viewRecv = nullptr;
    int32_t *viewCounts = nullptr;
    int32_t viewCapacity = 0;
    int32_t *lostDev = nullptr;            // vh_dist_raycast_auto: [0] this view's lost records, [1..R] every rank's
    int32_t autoCapacity = 0;              // ... and the slot capacity that last rendered every view whole
    int32_t autoStart = 4096;              // option "raycast_auto_start": this rank's proposal for the first round (the ranks take the largest)
    double hostSeconds = 0.0;
    uint64_t hostCalls = 0;
    // option "fused_generation": the key generation as a role of the frame launches (vh_shard.hip: GenJob)
    int fused = VH_DIST_FUSED_DEFAULT;     // 0 never, 1 where it pays (multi_fusing_pays: by the shard's size), 2 wherever the launch can carry it
    // ... and the form in effect: the fused host path when the shard's frames can carry the role at all (multi_can_fuse_generation:
    // not the walk-free launch, a band, float packets), else the separate path with its three streams overlapping as before.  Decided
    // where nothing is in flight and the streams are idle: at the first exchange and behind every vh_dist_flush.
    bool fusedActive = false, modeDirty = true;
    std::deque<int> inflight;              // fused path: the sets of the exchanges generated (and travelling / landed) but not applied yet: at most two
    bool genOnTable[kSets] = {};           // ... whose generation ran on the table stream (no event needed to apply it with one rank)
    bool headersClean[kSets] = {};         // ... whose send-bin headers the previous exchange's last job has zeroed
    hipEvent_t tableMark = nullptr;        // "everything queued on the table stream so far" (separate generation inside the fused path)
    bool forceCollectives = false;         // option "force_collectives": world 1 runs ncclAllToAll / ncclAllGather anyway (tests; the bench's first rounds)
    // option "phase_timing": per-exchange phase times from timing events of the library's own (vh_dist_phase_times)
    bool phaseTiming = false;
    struct PhaseEvents { hipEvent_t gen0 = nullptr, gen1 = nullptr, comm0 = nullptr, comm1 = nullptr, app0 = nullptr, app1 = nullptr; bool armed = false, applied = false; } phaseEv[kSets];
    vh_dist_phases phases{};
};

// ---------------------------------------------------------------------------
// transport entries
// ---------------------------------------------------------------------------
static int rccl_all_to_all(vh_dist *d, const void *send, void *recv, size_t bytes, hipStream_t s)
{
    if (bytes % 4 == 0) VH_RCCL(g_rccl.allToAll(send, recv, bytes / 4, ncclInt32, d->comm, s));
    else VH_RCCL(g_rccl.allToAll(send, recv, bytes, ncclUint8, d->comm, s));
    return VH_OK;
}
static int rccl_all_gather(vh_dist *d, const void *send, void *recv, size_t bytes, hipStream_t s)
{
    if (bytes % 4 == 0) VH_RCCL(g_rccl.allGather(send, recv, bytes / 4, ncclInt32, d->comm, s));
    else VH_RCCL(g_rccl.allGather(send, recv, bytes, ncclUint8, d->comm, s));
    return VH_OK;
}
static void rccl_leave(vh_dist *d)
{
    if (d->comm && d->ownComm && g_rccl.commDestroy) (void)g_rccl.commDestroy(d->comm);
    d->comm = nullptr;
}
static const Transport kRcclTransport = {"rccl", rccl_all_to_all, rccl_all_gather, rccl_leave};

// kind 0: all-to-all, 1: all-gather
static int loop_collective(vh_dist *d, int kind, const void *send, void *recv, size_t bytes, hipStream_t s)
{
    LoopGroup *g = d->loop.get();
    const int r = d->cfg.rank, R = d->cfg.world;
    LoopGroup::Post &mine = g->post[r];
    mine.send = send; mine.recv = recv; mine.bytes = bytes; mine.kind = kind;
    VH_HIP(hipEventRecord(mine.posted, s));
    if (!loop_meet(g)) return fail(VH_ERR_HIP, "loop-back transport: a rank of the group did not arrive at the collective (one host thread per rank is needed)");
    for (int i = 0; i < R; ++i) {
        const int p = (r + i) % R;                    // own part first, then the peers round the ring
        const LoopGroup::Post &peer = g->post[p];
        if (peer.kind != kind || peer.bytes != bytes) {
            { std::lock_guard<std::mutex> lk(g->m); g->broken = true; }
            g->cv.notify_all();
            return fail(VH_ERR_INVALID_ARGUMENT, "loop-back transport: the ranks disagree on the collective");
        }
        if (p != r) VH_HIP(hipStreamWaitEvent(s, peer.posted, 0));
        const uint8_t *src = static_cast<const uint8_t *>(peer.send) + (kind == 0 ? (size_t)r * bytes : 0);
        VH_HIP(hipMemcpyAsync(static_cast<uint8_t *>(recv) + (size_t)p * bytes, src, bytes, hipMemcpyDeviceToDevice, s));
    }
    VH_HIP(hipEventRecord(mine.done, s));
    if (!loop_meet(g)) return fail(VH_ERR_HIP, "loop-back transport: a rank of the group left the collective");
    for (int p = 0; p < R; ++p)
        if (p != r) VH_HIP(hipStreamWaitEvent(s, g->post[p].done, 0));
    return VH_OK;
}
static int loop_all_to_all(vh_dist *d, const void *send, void *recv, size_t bytes, hipStream_t s) { return loop_collective(d, 0, send, recv, bytes, s); }
static int loop_all_gather(vh_dist *d, const void *send, void *recv, size_t bytes, hipStream_t s) { return loop_collective(d, 1, send, recv, bytes, s); }
static void loop_leave(vh_dist *d)
{
    if (!d->loop) return;
    LoopGroup *g = d->loop.get();
    bool last = false;
    {
        std::lock_guard<std::mutex> lk(g->m);
        LoopGroup::Post &mine = g->post[d->cfg.rank];
        if (mine.posted) (void)hipEventDestroy(mine.posted);
        if (mine.done) (void)hipEventDestroy(mine.done);
        mine = LoopGroup::Post{};
        last = --g->joined == 0;
    }
    if (last) {
        std::lock_guard<std::mutex> lk(g_loop_mutex);
        g_loops.erase(g->serial);
    }
    d->loop.reset();
}
static const Transport kLoopTransport = {"loopback", loop_all_to_all, loop_all_gather, loop_leave};

static int loop_join(vh_dist *d, const char id[VH_DIST_ID_BYTES])
{
    uint64_t serial;
    std::memcpy(&serial, id + sizeof kLoopMagic, sizeof serial);
    std::shared_ptr<LoopGroup> g;
    {
        std::lock_guard<std::mutex> lk(g_loop_mutex);
        auto it = g_loops.find(serial);
        if (it == g_loops.end()) return fail(VH_ERR_INVALID_ARGUMENT, "loop-back id not issued by vh_dist_loopback_id of this process (or its group has been destroyed)");
        g = it->second;
    }
    std::lock_guard<std::mutex> lk(g->m);
    if (g->world == 0) g->world = d->cfg.world;
    if (g->world != d->cfg.world) return fail(VH_ERR_INVALID_ARGUMENT, "loop-back group: the ranks disagree on the world size");
    LoopGroup::Post &mine = g->post[d->cfg.rank];
    if (mine.present) return fail(VH_ERR_INVALID_ARGUMENT, "loop-back group: this rank has joined already");
    VH_HIP(hipEventCreateWithFlags(&mine.posted, hipEventDisableTiming));
    VH_HIP(hipEventCreateWithFlags(&mine.done, hipEventDisableTiming));
    mine.present = true;
    g->joined += 1;
    d->loop = g;
    return VH_OK;
}

static void dist_free(vh_dist *d)
{
    if (!d) return;
    DeviceGuard guard(d->device);
    (void)hipDeviceSynchronize();
    if (d->view) vh_destroy(d->view);
    if (d->shard) vh_destroy(d->shard);
    for (auto &s : d->set) {
        if (s.binsSend) (void)hipFree(s.binsSend);
        if (s.binsRecv) (void)hipFree(s.binsRecv);
        if (s.packet) (void)hipFree(s.packet);
        if (s.packets) (void)hipFree(s.packets);
    }
    for (void *p : {(void *)d->poseMine, (void *)d->poseAll, (void *)d->viewSend, (void *)d->viewRecv, (void *)d->viewCounts, (void *)d->lostDev})
        if (p) (void)hipFree(p);
    for (int i = 0; i < vh_dist::kSets; ++i)
        for (hipEvent_t e : {d->generated[i], d->ready[i], d->first[i]})
            if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : {d->userEvent, d->outEvent, d->tableMark})
        if (e) (void)hipEventDestroy(e);
    for (auto &pe : d->phaseEv)
        for (hipEvent_t e : {pe.gen0, pe.gen1, pe.comm0, pe.comm1, pe.app0, pe.app1})
            if (e) (void)hipEventDestroy(e);
    if (d->transport) d->transport->leave(d);
    for (hipStream_t s : {d->sGen, d->sComm, d->sTable})
        if (s) (void)hipStreamDestroy(s);
    delete d;
}

extern "C" int vh_dist_unique_id(char id[VH_DIST_ID_BYTES])
{
    if (!id) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    const int rc = rccl_load();
    if (rc != VH_OK) return rc;
    static_assert(sizeof(ncclUniqueId) == VH_DIST_ID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId u;
    VH_RCCL(g_rccl.getUniqueId(&u));
    std::memcpy(id, &u, sizeof u);
    return VH_OK;
}

extern "C" int vh_dist_loopback_id(char id[VH_DIST_ID_BYTES])
{
    if (!id) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> lk(g_loop_mutex);
    auto g = std::make_shared<LoopGroup>();
    g->serial = ++g_loop_serial;
    g_loops[g->serial] = g;
    std::memset(id, 0, VH_DIST_ID_BYTES);
    std::memcpy(id, kLoopMagic, sizeof kLoopMagic);
    std::memcpy(id + sizeof kLoopMagic, &g->serial, sizeof g->serial);
    return VH_OK;
}

extern "C" int vh_dist_probe(void)
{
    return rccl_load();
}

extern "C" const char *vh_dist_transport_name(vh_dist *d) { return d && d->transport ? d->transport->name : ""; }

extern "C" int vh_dist_create(const vh_dist_config *cfg, const char id[VH_DIST_ID_BYTES], void *nccl_comm, vh_dist **out)
{
    if (!cfg || !out || (!id && !nccl_comm)) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (cfg->world < 1 || cfg->world > VH_MAX_CAMERAS || cfg->rank < 0 || cfg->rank >= cfg->world || cfg->batch < 1 || cfg->batch > VH_MAX_CAMERAS)
        return fail(VH_ERR_INVALID_ARGUMENT, "bad rank / world / batch");
    if (cfg->packet_format != VH_PACKET_U16 && cfg->packet_format != VH_PACKET_F32)
        return fail(VH_ERR_INVALID_ARGUMENT, "bad packet format");
    const size_t npix = (size_t)cfg->table.width * cfg->table.height;
    if (cfg->packet_format == VH_PACKET_U16 && npix % 2) return fail(VH_ERR_INVALID_ARGUMENT, "sensor-depth packets need an even number of pixels");
    const bool loopback = !nccl_comm && std::memcmp(id, kLoopMagic, sizeof kLoopMagic) == 0;
    int rc = loopback ? VH_OK : rccl_load();
    if (rc != VH_OK) return rc;
    // bucket range of this rank: owner(h) = h / ceil(numBuckets / world)  (dist.py: ShardPlan)
    const uint32_t nb = cfg->table.params.numBuckets, R = (uint32_t)cfg->world;
    if (nb < R) return fail(VH_ERR_INVALID_ARGUMENT, "need at least one bucket per rank");
    const uint32_t per = (nb + R - 1) / R;
    const uint32_t lo = (uint32_t)cfg->rank * per, hi = std::min(nb, lo + per);
    if (lo >= hi) return fail(VH_ERR_INVALID_ARGUMENT, "this rank owns no bucket");

    vh_dist *d = new vh_dist();
    d->cfg = *cfg;
    if (const char *e = std::getenv("VOXELHASH_DIST_FUSED")) {           // the option's default (A/B and test switch, like VOXELHASH_LEAN_KERNELS)
        const int v = std::atoi(e);
        if (v >= 0 && v <= 2) d->fused = v;
    }
    rc = vh_create_shard(&cfg->table, lo, hi, &d->shard);
    if (rc != VH_OK) { dist_free(d); return rc; }
    d->device = d->shard->device;
    DeviceGuard guard(d->device);
    if (cfg->packet_format == VH_PACKET_U16) d->shard->packetFormat = VH_PACKET_U16;
    // one key bin per (owner, batch): 1.5 x the records a batch yields per owner when a frame yields W*H/16 (twice what the
    // 2-D wave dedup leaves of a room frame) and the owners share them evenly; bucket-range ownership is skewed by up to
    // ~2.5 x the even share on walls (the 8-rank rig), hence the factor on top.  Overflows are counted, never silent.
    const size_t perOwner = ((npix + 15) / 16 * (size_t)cfg->batch * 3 / 2 + (size_t)cfg->world - 1) / (size_t)cfg->world;
    // (the floor: small images split many ways have bins whose fullest exceeds 1.5 x the even share -- 320x240 over 3 or 8
    // ranks overflowed 4 801- and 2 048-record bins by 2 and 6 records in round 4's tests; 128 KB per bin costs nothing there)
    d->capacity = cfg->key_capacity > 0 ? cfg->key_capacity : (int)std::max<size_t>(8192, perOwner + 1);
    d->packetUnits = cfg->packet_format == VH_PACKET_U16 ? (size_t)kPacketHeaderU16 + npix / 2 : (size_t)kPacketHeader + npix;
    const size_t B = (size_t)cfg->batch;
    const size_t binBytes = (size_t)R * (size_t)d->capacity * 4 * sizeof(int32_t);      // [peer][capacity] records of 16 bytes
    const size_t pkBytes = B * d->packetUnits * sizeof(float);
#define VH_DIST_TRY(call)                                                                            \
    do {                                                                                             \
        const hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                      \
            dist_free(d);                                                                            \
            return fail(e_ == hipErrorOutOfMemory ? VH_ERR_OUT_OF_MEMORY : VH_ERR_HIP, #call, e_);   \
        }                                                                                            \
    } while (0)
    // (Round 4: the key-generation stream restricted to every 2nd / 4th / 8th CU with hipExtStreamCreateWithCUMask, so that the
    // generation of exchange n+1 would not slow the frame launch it runs beside (28 instead of 20 us): 43.9 / 43.4 / 43.2 k
    // frames/s against 43.8 k without a mask, the launch as slow as before -- what the two kernels contend for is not CUs.)
    for (hipStream_t *s : {&d->sGen, &d->sComm, &d->sTable}) VH_DIST_TRY(hipStreamCreateWithFlags(s, hipStreamNonBlocking));
    for (hipEvent_t *e : {&d->userEvent, &d->outEvent, &d->tableMark}) VH_DIST_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
    for (int i = 0; i < vh_dist::kSets; ++i) {
#ifndef VH_DIST_EVENT_FLAGS_FIRST
#define VH_DIST_EVENT_FLAGS_FIRST (hipEventDisableTiming | hipEventDisableSystemFence)
#endif
#ifndef VH_DIST_EVENT_FLAGS
#define VH_DIST_EVENT_FLAGS hipEventDisableTiming
#endif
        for (hipEvent_t *e : {&d->generated[i], &d->ready[i]}) VH_DIST_TRY(hipEventCreateWithFlags(e, VH_DIST_EVENT_FLAGS));
        // `first` only says that the frames are done READING a buffer set (it orders the next writer on this device behind them),
        // and it is recorded between two frame launches: no system-scope fence, whose cache write-back the next launch would pay
        VH_DIST_TRY(hipEventCreateWithFlags(&d->first[i], VH_DIST_EVENT_FLAGS_FIRST));
        VH_DIST_TRY(hipMalloc((void **)&d->set[i].binsSend, binBytes));
        VH_DIST_TRY(hipMalloc((void **)&d->set[i].binsRecv, binBytes));
        VH_DIST_TRY(hipMalloc((void **)&d->set[i].packet, pkBytes));
        VH_DIST_TRY(hipMalloc((void **)&d->set[i].packets, pkBytes * R));
        VH_DIST_TRY(hipMemset(d->set[i].binsSend, 0, binBytes));
        VH_DIST_TRY(hipMemset(d->set[i].binsRecv, 0, binBytes));
    }
    VH_DIST_TRY(hipDeviceSynchronize());
#undef VH_DIST_TRY
    if (loopback) {
        if ((rc = loop_join(d, id)) != VH_OK) { dist_free(d); return rc; }
        d->transport = &kLoopTransport;
    } else if (nccl_comm) {
        d->comm = reinterpret_cast<ncclComm_t>(nccl_comm);
        d->transport = &kRcclTransport;
    } else {
        ncclUniqueId u;
        std::memcpy(&u, id, sizeof u);
        const ncclResult_t r = g_rccl.commInitRank(&d->comm, cfg->world, u, cfg->rank);
        if (r != ncclSuccess) { dist_free(d); return rccl_fail("ncclCommInitRank", r); }
        d->ownComm = true;
        d->transport = &kRcclTransport;
    }
    d->shard->stream = d->sTable;
    d->shard->pipelineShards = 2;          // a batch's last frame rides in the first launch of the next batch
    *out = d;
    return VH_OK;
}

extern "C" int vh_dist_destroy(vh_dist *d)
{
    dist_free(d);
    return VH_OK;
}

extern "C" vh_context *vh_dist_shard(vh_dist *d) { return d ? d->shard : nullptr; }

extern "C" int vh_dist_set_user_stream(vh_dist *d, void *stream, int32_t enable)
{
    if (!d) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    d->userStream = (hipStream_t)stream;
    d->haveUser = enable != 0;
    return VH_OK;
}

// Phase timing (option "phase_timing"): six timing events per buffer set -- generation launch(es), the collectives, the frame
// launches of the set's exchange -- harvested when the set comes round again or at a flush.  Off by default: a timing event
// is a timestamp write behind a cache write-back, which the timed windows of bench.py do not pay.
static void phases_harvest(vh_dist *d, int s)
{
    vh_dist::PhaseEvents &e = d->phaseEv[s];
    if (!e.armed || !e.applied) return;
    // (the set comes round again three exchanges later: its frames are normally done; if not, this diagnostics mode waits)
    if (hipEventSynchronize(e.app1) != hipSuccess) return;
    float gen = 0, comm = 0, app = 0, lag = 0;
    if (hipEventElapsedTime(&gen, e.gen0, e.gen1) == hipSuccess && hipEventElapsedTime(&comm, e.comm0, e.comm1) == hipSuccess &&
        hipEventElapsedTime(&app, e.app0, e.app1) == hipSuccess && hipEventElapsedTime(&lag, e.gen0, e.app1) == hipSuccess) {
        d->phases.generate_us += 1e3 * gen; d->phases.collectives_us += 1e3 * comm; d->phases.apply_us += 1e3 * app;
        d->phases.first_to_last_us += 1e3 * lag;
        d->phases.exchanges += 1;
    }
    e.armed = false; e.applied = false;
}

extern "C" int vh_dist_set_option(vh_dist *d, const char *name, int32_t value)
{
    if (!d || !name) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    DeviceGuard guard(d->device);
    if (std::strcmp(name, "force_collectives") == 0) {
        if (d->pending >= 0 || !d->inflight.empty()) return fail(VH_ERR_INVALID_ARGUMENT, "force_collectives: set it before the first exchange or behind vh_dist_flush");
        d->forceCollectives = value != 0;
        return VH_OK;
    }
    if (std::strcmp(name, "raycast_auto_start") == 0 && value >= 1) { d->autoStart = value; return VH_OK; }
    if (std::strcmp(name, "fused_generation") == 0) {
        if (d->pending >= 0 || !d->inflight.empty()) return fail(VH_ERR_INVALID_ARGUMENT, "fused_generation: set it before the first exchange or behind vh_dist_flush");
        if (value < 0 || value > 2) return fail(VH_ERR_INVALID_ARGUMENT, "fused_generation: 0, 1 or 2");
        d->fused = value;
        d->modeDirty = true;                  // (vh_dist_step_batch decides the form in effect; nothing is in flight)
        return VH_OK;
    }
    if (std::strcmp(name, "phase_timing") == 0) {
        if (value && !d->phaseEv[0].gen0)
            for (auto &e : d->phaseEv)
                for (hipEvent_t *ev : {&e.gen0, &e.gen1, &e.comm0, &e.comm1, &e.app0, &e.app1}) VH_HIP(hipEventCreate(ev));
        for (auto &e : d->phaseEv) { e.armed = false; e.applied = false; }
        d->phaseTiming = value != 0;
        return VH_OK;
    }
    return fail(VH_ERR_INVALID_ARGUMENT, "unknown option");
}

extern "C" int vh_dist_phase_times(vh_dist *d, vh_dist_phases *out, int32_t reset)
{
    if (!d || !out) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    DeviceGuard guard(d->device);
    for (int s = 0; s < vh_dist::kSets; ++s) phases_harvest(d, s);
    *out = d->phases;
    out->host_enqueue_us = d->hostCalls ? 1e6 * d->hostSeconds / (double)d->hostCalls * (double)d->phases.exchanges : 0.0;
    if (reset) d->phases = vh_dist_phases{};
    return VH_OK;
}

// Start-up self-check of the transport (VERDICT round 4, next #4c): a known pattern through the same two collectives an exchange
// uses -- all-to-all of one 64 KB slice per peer, all-gather of 64 KB -- compared on the device, before anything is timed or fused.
__global__ void dist_pattern_kernel(uint32_t *a2a, uint32_t *gather, int32_t rank, int32_t world, uint32_t words)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= words) return;
    for (int p = 0; p < world; ++p) a2a[(size_t)p * words + i] = 0x9e3779b9u * (uint32_t)(rank * 64 + p + 1) + i;     // what `rank` sends to p
    gather[i] = 0x85ebca6bu * (uint32_t)(rank + 1) ^ i;
}
__global__ void dist_pattern_check_kernel(const uint32_t *a2a, const uint32_t *gather, int32_t rank, int32_t world, uint32_t words, int32_t *bad)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= words) return;
    int n = 0;
    for (int p = 0; p < world; ++p) {
        n += a2a[(size_t)p * words + i] != 0x9e3779b9u * (uint32_t)(p * 64 + rank + 1) + i;      // what p sent to `rank`
        n += gather[(size_t)p * words + i] != (0x85ebca6bu * (uint32_t)(p + 1) ^ i);
    }
    if (n) atomicAdd(bad, n);
}
// Drain + synchronise.  late != nullptr: the caller is about to enter collectives -- a rank-local, non-fatal condition
// (VH_ERR_TIMEOUT: a serialised launch of THIS shard gave up waiting) must not make this rank leave the round while its peers
// enter ncclAllGather without it: it is latched into *late, VH_OK is returned, and the caller returns *late
// after the round's last collective.
static int dist_flush_sync(vh_dist *d, int *late);

extern "C" int vh_dist_self_check(vh_dist *d)
{
    if (!d) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    DeviceGuard guard(d->device);
    int late = VH_OK;
    int rc = dist_flush_sync(d, &late);
    if (rc != VH_OK) return rc;
    const int R = d->cfg.world;
    // up to 64 KB per peer, what the smaller of a key bin and this rank's packets holds (in whole 1 KB pieces)
    const uint32_t words = (uint32_t)std::min<size_t>(16384, std::min<size_t>((size_t)d->capacity * 4, (size_t)d->cfg.batch * d->packetUnits)) & ~255u;
    if (words == 0) return fail(VH_ERR_INVALID_ARGUMENT, "vh_dist_self_check: exchange buffers smaller than 1 KB");
    vh_dist::Set &set = d->set[0];
    int32_t *bad = nullptr;
    VH_HIP(hipMalloc((void **)&bad, sizeof(int32_t)));
    VH_HIP(hipMemsetAsync(bad, 0, sizeof(int32_t), d->sComm));
    dist_pattern_kernel<<<dim3(words / 256), dim3(256), 0, d->sComm>>>(reinterpret_cast<uint32_t *>(set.binsSend), reinterpret_cast<uint32_t *>(set.packet), d->cfg.rank, R, words);
    rc = d->transport->all_to_all(d, set.binsSend, set.binsRecv, (size_t)words * 4, d->sComm);
    if (rc == VH_OK) rc = d->transport->all_gather(d, set.packet, set.packets, (size_t)words * 4, d->sComm);
    if (rc != VH_OK) { (void)hipFree(bad); return rc; }
    dist_pattern_check_kernel<<<dim3(words / 256), dim3(256), 0, d->sComm>>>(reinterpret_cast<const uint32_t *>(set.binsRecv), reinterpret_cast<const uint32_t *>(set.packets), d->cfg.rank, R, words, bad);
    int32_t n = -1;
    hipError_t e = hipMemcpyAsync(&n, bad, sizeof n, hipMemcpyDeviceToHost, d->sComm);
    if (e == hipSuccess) e = hipStreamSynchronize(d->sComm);
    // the buffers go back to what vh_dist_create left: empty bins (a header record of zero keys)
    const size_t binBytes = (size_t)R * (size_t)d->capacity * 4 * sizeof(int32_t);
    if (e == hipSuccess) e = hipMemsetAsync(set.binsSend, 0, binBytes, d->sComm);
    if (e == hipSuccess) e = hipMemsetAsync(set.binsRecv, 0, binBytes, d->sComm);
    if (e == hipSuccess) e = hipStreamSynchronize(d->sComm);
    (void)hipFree(bad);
    if (e != hipSuccess) return fail(VH_ERR_HIP, "vh_dist_self_check", e);
    if (n == 0 && late != VH_OK) return late;
    if (n != 0) {
        char msg[160];
        std::snprintf(msg, sizeof msg, "vh_dist_self_check: rank %d of %d received %d wrong words through the %s transport", d->cfg.rank, R, n, d->transport->name);
        return fail(VH_ERR_HIP, msg);
    }
    return VH_OK;
}

// jobs (fused path, nullable): the generation jobs that ride in this exchange's frame launches
static int dist_apply(vh_dist *d, int s, const GenJob *jobs = nullptr)
{
    const bool alone = d->cfg.world == 1 && !d->forceCollectives;
#ifndef VH_DEBUG_DIST_NO_READY_WAIT          // (diagnostics builds: what the two event operations at a batch's boundary cost)
    // (fused path, one rank: the exchange was generated on this very stream and nothing travelled: stream order is the hand-off)
    if (!(d->fusedActive && alone && d->genOnTable[s])) VH_HIP(hipStreamWaitEvent(d->sTable, d->ready[s], 0));
#endif
    d->shard->stream = d->sTable;
#ifndef VH_DEBUG_DIST_NO_FIRST
    if (!d->fusedActive) d->shard->multiFirstEvent = d->first[s];
#endif
    if (d->phaseTiming && d->phaseEv[s].armed) VH_HIP(hipEventRecord(d->phaseEv[s].app0, d->sTable));
    const int rc = vh_apply_frames_batch_gen(d->shard, d->cfg.batch, alone ? d->set[s].binsSend : d->set[s].binsRecv, d->cfg.world, d->capacity, 0,
                                             VH_BIN_PER_BATCH, d->cfg.world, alone ? d->set[s].packet : d->set[s].packets, 0, 0, jobs);
    d->shard->multiFirstEvent = nullptr;
    if (rc == VH_OK && d->phaseTiming && d->phaseEv[s].armed) {
        VH_HIP(hipEventRecord(d->phaseEv[s].app1, d->sTable));
        d->phaseEv[s].applied = true;
    }
    return rc;
}

// The fused path (option "fused_generation", the default): exchange n is GENERATED inside the frame launches that apply exchange
// n-2 (GenJob: a role of frame_multi_pipelined_kernel), its collectives travel beside the launches of exchange n-1, and it is applied
// in call n+2 with the generation of n+2 aboard.  One stream carries table work and generation, so nothing of the library runs
// beside the frame launches: with one rank a steady-state call is B launches and NO event operation.  Whenever a call cannot fuse
// -- the first two calls (nothing to ride in), a band, float packets, the overflow list -- it generates with the kernels of the
// separate path on sGen, ordered behind "everything queued on the table stream so far".
static int dist_step_fused(vh_dist *d, const float *poses, const void *const *d_frames)
{
    const auto t0 = std::chrono::steady_clock::now();
    DeviceGuard guard(d->device);
    const int s = (int)(d->count % vh_dist::kSets), sNext = (s + 1) % vh_dist::kSets;
    const int B = d->cfg.batch, R = d->cfg.world;
    vh_dist::Set &set = d->set[s];
    const bool alone = R == 1 && !d->forceCollectives;
    int rc;
    const bool canFuse = d->inflight.size() == 2 && d->cfg.packet_format == VH_PACKET_U16 && B <= 8 && d->capacity >= 8 &&
                         multi_can_fuse_generation(d->shard, R, d->capacity);
    if (d->haveUser) VH_HIP(hipEventRecord(d->userEvent, d->userStream));
    if (d->phaseTiming) {
        phases_harvest(d, s);
        d->phaseEv[s].armed = hipEventRecord(d->phaseEv[s].gen0, canFuse ? d->sTable : d->sGen) == hipSuccess;
        d->phaseEv[s].applied = false;
    }
    GenJob jobs[VH_MAX_CAMERAS];
    if (canFuse) {
        const uint32_t tiles = host_num_tiles(d->shard);
        for (int b = 0; b < B; ++b) {
            GenJob &j = jobs[b];
            std::memset(&j, 0, sizeof j);
            if ((rc = vh_set_pose(d->shard, poses + 16 * (size_t)b)) != VH_OK) return rc;          // pose + cofactor inverse
            if (!d_frames[b]) return fail(VH_ERR_INVALID_ARGUMENT, "null depth image");
            std::memcpy(j.T, d->shard->fp.T, sizeof j.T);
            std::memcpy(j.Tinv, d->shard->fp.Tinv, sizeof j.Tinv);
            std::memcpy(j.k, d->cfg.k_inv, sizeof j.k);
            j.unit = 5000.0f;                                                                       // CameraTrackingUtils.cu:64
            j.blocks = (tiles + kFusedGenGroups - 1) / kFusedGenGroups;
            j.numShards = R; j.capacity = d->capacity; j.binStride = d->capacity;
            j.bins = reinterpret_cast<int4 *>(set.binsSend);
            j.packet = set.packet + (size_t)b * d->packetUnits;
            j.depth = reinterpret_cast<const uint16_t *>(d_frames[b]);
            j.rankBase = (uint32_t)b << kRankCameraShift;
            if (b == B - 1) { j.clearBins = reinterpret_cast<int4 *>(d->set[sNext].binsSend); j.clearStride = d->capacity; }
            j.frame = b; j.batch = B;
        }
    } else {
        // separate generation: this set's buffers were last read by table work queued in earlier calls
        VH_HIP(hipEventRecord(d->tableMark, d->sTable));
        VH_HIP(hipStreamWaitEvent(d->sGen, d->tableMark, 0));
        if (d->haveUser) VH_HIP(hipStreamWaitEvent(d->sGen, d->userEvent, 0));
        d->shard->stream = d->sGen;
        if (d->cfg.packet_format == VH_PACKET_U16)
            rc = vh_generate_keys_depth_batch(d->shard, B, poses, reinterpret_cast<const uint16_t *const *>(d_frames), d->cfg.k_inv,
                                              (uint32_t)d->cfg.rank, R, set.binsSend, d->capacity, 0, VH_BIN_PER_BATCH, set.packet, 0);
        else
            rc = vh_generate_keys_batch(d->shard, B, poses, reinterpret_cast<const vh_float4 *const *>(d_frames), (uint32_t)d->cfg.rank, R,
                                        set.binsSend, d->capacity, 0, VH_BIN_PER_BATCH, set.packet, 0);
        d->shard->stream = d->sTable;
        if (rc != VH_OK) return rc;
        if (d->phaseTiming && d->phaseEv[s].armed) VH_HIP(hipEventRecord(d->phaseEv[s].gen1, d->sGen));
        VH_HIP(hipEventRecord(d->generated[s], d->sGen));
        d->genOnTable[s] = false;
    }
    // apply exchange n-2 (with this exchange's generation aboard when it can ride)
    bool quiet = false;                     // one rank, fused, nobody else to tell: no event operation at all
    if (d->inflight.size() == 2) {
        const int old = d->inflight.front();
        d->inflight.pop_front();
        if (canFuse) {
            if (d->haveUser) VH_HIP(hipStreamWaitEvent(d->sTable, d->userEvent, 0));
            if (!d->headersClean[s]) {      // (the first fused call behind a separate one: nobody has emptied this set's send bins)
                prepare_bins_fused_kernel<<<1, 64, 0, d->sTable>>>(reinterpret_cast<int4 *>(set.binsSend), R, d->capacity, d->capacity, B);
            }
        }
        if ((rc = dist_apply(d, old, canFuse ? jobs : nullptr)) != VH_OK) return rc;
        if (canFuse) {
            d->genOnTable[s] = true;
            d->headersClean[sNext] = true;
            quiet = alone && !d->haveUser && !d->phaseTiming;
            if (!quiet) {
                if (d->phaseTiming && d->phaseEv[s].armed) VH_HIP(hipEventRecord(d->phaseEv[s].gen1, d->sTable));
                VH_HIP(hipEventRecord(d->generated[s], d->sTable));
            }
        }
    }
    d->headersClean[s] = false;
    if (!quiet) {
        // (the frames have been consumed once `generated` fires: the caller's stream may overwrite them behind it)
        if (d->haveUser) VH_HIP(hipStreamWaitEvent(d->userStream, d->generated[s], 0));
        // exchange: the receive buffers of this set were last read by table work of earlier calls -- behind which `generated` lies when
        // it was recorded on the table stream, and behind tableMark otherwise
        VH_HIP(hipStreamWaitEvent(d->sComm, d->generated[s], 0));
        if (!canFuse) VH_HIP(hipStreamWaitEvent(d->sComm, d->tableMark, 0));
        if (d->phaseTiming && d->phaseEv[s].armed) VH_HIP(hipEventRecord(d->phaseEv[s].comm0, d->sComm));
        if (!alone) {
            if ((rc = d->transport->all_to_all(d, set.binsSend, set.binsRecv, (size_t)d->capacity * 4 * sizeof(int32_t), d->sComm)) != VH_OK) return rc;
            if ((rc = d->transport->all_gather(d, set.packet, set.packets, (size_t)B * d->packetUnits * sizeof(float), d->sComm)) != VH_OK) return rc;
        }
        if (d->phaseTiming && d->phaseEv[s].armed) VH_HIP(hipEventRecord(d->phaseEv[s].comm1, d->sComm));
        VH_HIP(hipEventRecord(d->ready[s], d->sComm));
    }
    d->inflight.push_back(s);
    d->count += 1;
    d->hostSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    d->hostCalls += 1;
    return VH_OK;
}

extern "C" int vh_dist_step_batch(vh_dist *d, const float *poses, const void *const *d_frames)
{
    VH_TRACE("vh_dist_step_batch");
    if (!d || !poses || !d_frames) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    const auto t0 = std::chrono::steady_clock::now();
    DeviceGuard guard(d->device);
    if (d->modeDirty) {
        const bool want = d->fused != 0 && d->cfg.packet_format == VH_PACKET_U16 && d->cfg.batch <= 8 && d->capacity >= 8 &&
                          multi_can_fuse_generation(d->shard, d->cfg.world, d->capacity) && (d->fused == 2 || multi_fusing_pays(d->shard));
        if (want != d->fusedActive) {
            d->fusedActive = want;
            d->count = 0;                     // (the set rotation restarts: nothing is in flight, the streams are idle)
            for (bool &b : d->headersClean) b = false;
        }
        d->modeDirty = false;
    }
    if (d->fusedActive) return dist_step_fused(d, poses, d_frames);
    const int s = (int)(d->count % vh_dist::kSetsSeparate);
    const int B = d->cfg.batch, R = d->cfg.world;
    vh_dist::Set &set = d->set[s];
    int rc;
    // generate: keys binned by owner + this camera's packets.  The set's send buffers were last read by the collectives
    // of exchange count-3.  The frames are read behind whatever the caller's stream has queued (vh_dist_set_user_stream).
    if (d->haveUser) {
        VH_HIP(hipEventRecord(d->userEvent, d->userStream));
        VH_HIP(hipStreamWaitEvent(d->sGen, d->userEvent, 0));
    }
    if (d->count >= vh_dist::kSetsSeparate) VH_HIP(hipStreamWaitEvent(d->sGen, d->ready[s], 0));
    // (one rank: the frames are applied straight from the send buffers, so those are free only when the frames of exchange
    // count-3 are done, the last of which rode in the first launch of exchange count-2's frames)
    const bool alone = R == 1 && !d->forceCollectives;
    if (alone && d->count >= vh_dist::kSetsSeparate) VH_HIP(hipStreamWaitEvent(d->sGen, d->first[(s + 1) % vh_dist::kSetsSeparate], 0));
    if (d->phaseTiming) {
        phases_harvest(d, s);
        d->phaseEv[s].armed = hipEventRecord(d->phaseEv[s].gen0, d->sGen) == hipSuccess;
        d->phaseEv[s].applied = false;
    }
    d->shard->stream = d->sGen;
#ifdef VH_DEBUG_DIST_SKIP_GEN
    // diagnostics build (tools/ab_variants.sh): from the 7th exchange on nothing is generated -- the frames re-apply what the
    // buffer set holds -- so that the frame launches can be timed without a key generation beside them
    if (d->count >= 6) rc = VH_OK;
    else
#endif
    if (d->cfg.packet_format == VH_PACKET_U16)
        rc = vh_generate_keys_depth_batch(d->shard, B, poses, reinterpret_cast<const uint16_t *const *>(d_frames), d->cfg.k_inv,
                                          (uint32_t)d->cfg.rank, R, set.binsSend, d->capacity, 0, VH_BIN_PER_BATCH, set.packet, 0);
    else
        rc = vh_generate_keys_batch(d->shard, B, poses, reinterpret_cast<const vh_float4 *const *>(d_frames), (uint32_t)d->cfg.rank, R,
                                    set.binsSend, d->capacity, 0, VH_BIN_PER_BATCH, set.packet, 0);
    d->shard->stream = d->sTable;
    if (rc != VH_OK) return rc;
    if (d->phaseTiming && d->phaseEv[s].armed) VH_HIP(hipEventRecord(d->phaseEv[s].gen1, d->sGen));
    VH_HIP(hipEventRecord(d->generated[s], d->sGen));
    // (the frames have been consumed once `generated` fires: the caller's stream may overwrite them behind it)
    if (d->haveUser) VH_HIP(hipStreamWaitEvent(d->userStream, d->generated[s], 0));
    // exchange: the receive buffers of this set were last read by the frames of exchange count-3, the last of which rode in
    // the first launch of exchange count-2's frames (queued by the previous call)
    VH_HIP(hipStreamWaitEvent(d->sComm, d->generated[s], 0));
    if (d->count >= vh_dist::kSetsSeparate) VH_HIP(hipStreamWaitEvent(d->sComm, d->first[(s + 1) % vh_dist::kSetsSeparate], 0));
    if (d->phaseTiming && d->phaseEv[s].armed) VH_HIP(hipEventRecord(d->phaseEv[s].comm0, d->sComm));      // (behind the waits: when the collectives may start)
    if (!alone) {
        if ((rc = d->transport->all_to_all(d, set.binsSend, set.binsRecv, (size_t)d->capacity * 4 * sizeof(int32_t), d->sComm)) != VH_OK) return rc;
        if ((rc = d->transport->all_gather(d, set.packet, set.packets, (size_t)B * d->packetUnits * sizeof(float), d->sComm)) != VH_OK) return rc;
    }
    if (d->phaseTiming && d->phaseEv[s].armed) VH_HIP(hipEventRecord(d->phaseEv[s].comm1, d->sComm));
    // (one rank: the only bin and the only packet are this rank's own -- the frames are applied straight from the send
    // buffers, no collective and no copy; the events order the hand-offs as with peers)
    VH_HIP(hipEventRecord(d->ready[s], d->sComm));
    // apply the previous exchange while this one travels
    if (d->pending >= 0 && (rc = dist_apply(d, d->pending)) != VH_OK) return rc;
    d->pending = s;
    d->count += 1;
    d->hostSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    d->hostCalls += 1;
    return VH_OK;
}

// applies the exchange in flight and the last frame's deferred half: everything fed is queued on the table stream
static int dist_drain(vh_dist *d)
{
    while (!d->inflight.empty()) {          // fused path: up to two exchanges generated, not applied
        const int s = d->inflight.front();
        d->inflight.pop_front();
        const int rc = dist_apply(d, s);
        if (rc != VH_OK) return rc;
    }
    if (d->pending >= 0) {
        const int rc = dist_apply(d, d->pending);
        if (rc != VH_OK) return rc;
        d->pending = -1;
    }
    return flush_pending(d->shard);
}

static int dist_flush_sync(vh_dist *d, int *late)
{
    const int rc = dist_drain(d);
    if (rc != VH_OK) return rc;
    VH_HIP(hipStreamSynchronize(d->sTable));
    VH_HIP(hipStreamSynchronize(d->sComm));
    VH_HIP(hipStreamSynchronize(d->sGen));
    d->modeDirty = true;
    const int t = check_spin_timeouts(d->shard);      // (VH_ERR_TIMEOUT: a serialised multi-camera launch gave up waiting, voxelhash.h)
    if (t == VH_ERR_TIMEOUT && late) {
        *late = t;
        return VH_OK;
    }
    return t;
}

extern "C" int vh_dist_flush(vh_dist *d)
{
    if (!d) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    DeviceGuard guard(d->device);
    return dist_flush_sync(d, nullptr);
}

static int dist_raycast_impl(vh_dist *d, const float pose[16], float t_min, float t_max, int32_t capacity, float *d_depth_out,
                             vh_float4 *d_normals_out, int32_t *d_lost)
{
    VH_TRACE("vh_dist_raycast");
    if (!d || !pose || !d_depth_out || capacity < 1) return fail(VH_ERR_INVALID_ARGUMENT, "bad argument");
    const int R = d->cfg.world;
    if (R > 16) return fail(VH_ERR_INVALID_ARGUMENT, "the fixed-slot raycast round serves at most 16 views");
    DeviceGuard guard(d->device);
    int late = VH_OK;                               // (a spin timeout of this shard: reported after the round's collectives)
    int rc = dist_flush_sync(d, &late);             // the model as of every fed exchange
    if (rc != VH_OK) return rc;
    if (!d->view) {
        vh_config vc = d->cfg.table;
        vc.params.numVoxelBlocks = 1;               // the voxels of a view table stay in the received records
        vc.device = d->device;
        if ((rc = vh_create(&vc, &d->view)) != VH_OK) return rc;
        d->view->stream = d->sTable;
        VH_HIP(hipMalloc((void **)&d->poseMine, 16 * sizeof(float)));
        VH_HIP(hipMalloc((void **)&d->poseAll, (size_t)R * 16 * sizeof(float)));
        VH_HIP(hipMalloc((void **)&d->viewCounts, (size_t)R * sizeof(int32_t)));
    }
    // the view table renders with the shard's raycast settings as they are NOW (vh_set_option / vh_set_raycast_intrinsics
    // on vh_dist_shard() between rounds take effect)
    d->view->rc_fx = d->shard->rc_fx; d->view->rc_fy = d->shard->rc_fy; d->view->rc_cx = d->shard->rc_cx; d->view->rc_cy = d->shard->rc_cy;
    d->view->raycastMode = d->shard->raycastMode;
    d->view->raycastBeam = d->shard->raycastBeam;
    d->view->fp.flags = (d->view->fp.flags & ~kFlagOverflow) | (d->shard->fp.flags & kFlagOverflow);
    d->view->fp.listSize = d->shard->fp.listSize;
    if (d->viewCapacity < capacity) {
        VH_HIP(hipStreamSynchronize(d->sTable));
        if (d->viewSend) (void)hipFree(d->viewSend);
        if (d->viewRecv) (void)hipFree(d->viewRecv);
        d->viewSend = d->viewRecv = nullptr;
        d->viewCapacity = 0;
        const size_t bytes = (size_t)R * capacity * sizeof(vh_view_record);
        VH_HIP(hipMalloc((void **)&d->viewSend, bytes));
        VH_HIP(hipMalloc((void **)&d->viewRecv, bytes));
        VH_HIP(hipMemsetAsync(d->viewSend, 0, bytes, d->sTable));
        VH_HIP(hipMemsetAsync(d->viewRecv, 0, bytes, d->sTable));
        d->viewCapacity = capacity;
    }
    // the image (and d_lost) are written behind whatever the caller's stream has queued, and the caller's stream reads them
    // behind the raycast (vh_dist_set_user_stream)
    if (d->haveUser) {
        VH_HIP(hipEventRecord(d->userEvent, d->userStream));
        VH_HIP(hipStreamWaitEvent(d->sTable, d->userEvent, 0));
    }
    Pose16 p;
    std::memcpy(p.m, pose, sizeof p.m);
    dist_store_pose_kernel<<<1, 64, 0, d->sTable>>>(p, d->poseMine);
    if ((rc = d->transport->all_gather(d, d->poseMine, d->poseAll, 16 * sizeof(float), d->sTable)) != VH_OK) return rc;
    d->shard->stream = d->sTable;
    if ((rc = vh_export_views_fixed(d->shard, d->poseAll, R, t_min, t_max, d->viewSend, capacity, d->viewCounts)) != VH_OK) return rc;
    if ((rc = d->transport->all_to_all(d, d->viewSend, d->viewRecv, (size_t)capacity * sizeof(vh_view_record), d->sTable)) != VH_OK) return rc;
    if ((rc = vh_import_views(d->view, d->viewRecv, R, capacity, nullptr)) != VH_OK) return rc;
    rc = d_normals_out ? vh_raycast_normals(d->view, pose, t_min, t_max, d_depth_out, d_normals_out)
                       : vh_raycast(d->view, pose, t_min, t_max, d_depth_out);
    if (rc != VH_OK) return rc;
    if (d_lost) dist_lost_kernel<<<1, 64, 0, d->sTable>>>(reinterpret_cast<const uint8_t *>(d->viewRecv), R, capacity, d_lost);
    VH_HIP(hipGetLastError());
    if (d->haveUser) {
        VH_HIP(hipEventRecord(d->outEvent, d->sTable));
        VH_HIP(hipStreamWaitEvent(d->userStream, d->outEvent, 0));
    }
    return late;
}

extern "C" int vh_dist_raycast(vh_dist *d, const float pose[16], float t_min, float t_max, int32_t capacity, float *d_depth_out,
                               int32_t *d_lost)
{
    return dist_raycast_impl(d, pose, t_min, t_max, capacity, d_depth_out, nullptr, d_lost);
}

// The round with the slot capacity found by the library: rendered, the ranks' lost counts gathered (a view with holes on any
// rank repeats the round for all, since the slots are one size everywhere), repeated with room for what was lost.
extern "C" int vh_dist_raycast_auto(vh_dist *d, const float pose[16], float t_min, float t_max, float *d_depth_out,
                                    vh_float4 *d_normals_out, int32_t *capacity_used)
{
    if (!d) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    DeviceGuard guard(d->device);
    const int R = d->cfg.world;
    if (!d->lostDev) {
        VH_HIP(hipMalloc((void **)&d->lostDev, sizeof(int32_t) * (size_t)(R + 1)));
        VH_HIP(hipMemset(d->lostDev, 0, sizeof(int32_t) * (size_t)(R + 1)));
    }
    // a shard cannot select more blocks than its pool holds, and the view table lists one imported record per entry
    const int64_t most = std::max<int64_t>(1, std::min<int64_t>((int64_t)d->cfg.table.params.numVoxelBlocks,
        (int64_t)d->cfg.table.params.numBuckets * d->cfg.table.params.bucketSize / R));
    int late = VH_OK;                                   // a rank-local spin timeout: every collective of the call still runs, then it is returned
    const int32_t first = d->autoStart;                 // (option "raycast_auto_start": 4096 unless a caller -- the test of the retry -- asks otherwise)
    int32_t cap = (int32_t)std::min<int64_t>(most, std::max<int32_t>(d->autoCapacity, first));
    {
        // The slots are one size everywhere and every rank issues an all-to-all of that size: the ranks AGREE on the first
        // capacity (the largest any of them proposes) instead of trusting that their local state -- the capacity of their last
        // call, an environment variable -- is identical; ranks that disagreed would hang in mismatched collectives.
        int rc = dist_flush_sync(d, &late);
        if (rc != VH_OK) return rc;
        VH_HIP(hipMemcpyAsync(d->lostDev, &cap, sizeof cap, hipMemcpyHostToDevice, d->sTable));
        if ((rc = d->transport->all_gather(d, d->lostDev, d->lostDev + 1, sizeof(int32_t), d->sTable)) != VH_OK) return rc;
        int32_t proposed[VH_MAX_CAMERAS];
        VH_HIP(hipMemcpyAsync(proposed, d->lostDev + 1, sizeof(int32_t) * (size_t)R, hipMemcpyDeviceToHost, d->sTable));
        VH_HIP(hipStreamSynchronize(d->sTable));
        for (int r = 0; r < R; ++r) cap = std::max(cap, proposed[r]);
        cap = (int32_t)std::min<int64_t>(most, cap);
    }
    const int kAttempts = 8;
    for (int attempt = 0; attempt < kAttempts; ++attempt) {
        int rc = dist_raycast_impl(d, pose, t_min, t_max, cap, d_depth_out, d_normals_out, d->lostDev);
        if (rc == VH_ERR_TIMEOUT) { late = rc; rc = VH_OK; }
        if (rc != VH_OK) return rc;
        if ((rc = d->transport->all_gather(d, d->lostDev, d->lostDev + 1, sizeof(int32_t), d->sTable)) != VH_OK) return rc;
        int32_t lost[VH_MAX_CAMERAS];
        VH_HIP(hipMemcpyAsync(lost, d->lostDev + 1, sizeof(int32_t) * (size_t)R, hipMemcpyDeviceToHost, d->sTable));
        VH_HIP(hipStreamSynchronize(d->sTable));
        int32_t worst = 0;
        for (int r = 0; r < R; ++r) worst = std::max(worst, lost[r]);
        if (worst == 0) {
            d->autoCapacity = cap;
            if (capacity_used) *capacity_used = cap;
            return late;
        }
        if ((int64_t)cap >= most)
            return fail(VH_ERR_INVALID_ARGUMENT, "vh_dist_raycast_auto: a view selects more blocks of a shard than the shard's pool holds");
        cap = (int32_t)std::min<int64_t>(most, std::max<int64_t>((int64_t)cap + worst, (int64_t)cap * 2));
    }
    return fail(VH_ERR_INVALID_ARGUMENT, "vh_dist_raycast_auto: views still lost records after 8 rounds of growing the slots (the capacity at least doubles "
                                         "every round): start larger (a previous call's capacity_used) or use vh_dist_raycast with a capacity of your own");
}

extern "C" int vh_dist_comm_info(vh_dist *d, int32_t *rank, int32_t *world)
{
    if (!d) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    int r = d->cfg.rank, n = d->cfg.world;
    if (d->transport == &kRcclTransport) {           // what the communicator itself says
        VH_RCCL(g_rccl.commCount(d->comm, &n));
        VH_RCCL(g_rccl.commUserRank(d->comm, &r));
    } else if (d->loop) {
        std::lock_guard<std::mutex> lk(d->loop->m);
        n = d->loop->joined;
    }
    if (rank) *rank = r;
    if (world) *world = n;
    return VH_OK;
}

extern "C" int vh_dist_generation_form(vh_dist *d)
{
    if (!d) return -fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    if (d->modeDirty) {                  // (no exchange since creation / the last flush: what the next one will decide)
        const bool want = d->fused != 0 && d->cfg.packet_format == VH_PACKET_U16 && d->cfg.batch <= 8 && d->capacity >= 8 &&
                          multi_can_fuse_generation(d->shard, d->cfg.world, d->capacity) && (d->fused == 2 || multi_fusing_pays(d->shard));
        return want ? 1 : 0;
    }
    return d->fusedActive ? 1 : 0;
}

extern "C" int vh_dist_host_stats(vh_dist *d, double *seconds, uint64_t *calls)
{
    if (!d) return fail(VH_ERR_INVALID_ARGUMENT, "null argument");
    if (seconds) *seconds = d->hostSeconds;
    if (calls) *calls = d->hostCalls;
    return VH_OK;
}
