// A C++ host of the sharded table with R > 1 ranks on ONE GPU: R SDF_Hashtable ranks of this process (multi-GPU constructor,
// include/SDF_Hashtable.h), one std::thread per rank -- what R processes on R GPUs do, with the loop-back transport
// (SDF_Hashtable::loopbackId) carrying the exchange instead of RCCL.  Every rank feeds `steps` exchanges of `batch` uint16 sensor
// frames of ITS camera, flushes, dumps its shard's hash table and renders its own view through all shards.  No Python, no torch.
//   sharded_threads_demo R poses.bin depth.bin kinv.bin W H batch steps numBuckets numVoxelBlocks out_prefix
//   poses.bin: [R][steps*batch][16] floats, depth.bin: [R][steps*batch][H][W] uint16; writes <out_prefix>table<r>.bin, depth<r>.bin
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "SDF_Hashtable.h"

template <class T>
static bool read_all(const char *path, std::vector<T> &v)
{
    FILE *f = std::fopen(path, "rb");
    if (!f) return false;
    const bool ok = std::fread(v.data(), sizeof(T), v.size(), f) == v.size();
    std::fclose(f);
    return ok;
}

template <class T>
static bool write_all(const std::string &path, const std::vector<T> &v)
{
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool ok = std::fwrite(v.data(), sizeof(T), v.size(), f) == v.size();
    std::fclose(f);
    return ok;
}

int main(int argc, char **argv)
{
    if (argc < 12) return 2;
    const int R = std::atoi(argv[1]);
    const int W = std::atoi(argv[5]), H = std::atoi(argv[6]), batch = std::atoi(argv[7]), steps = std::atoi(argv[8]);
    const size_t npix = (size_t)W * H, nframes = (size_t)batch * steps;
    std::vector<float> poses((size_t)R * nframes * 16), kinv(9);
    std::vector<uint16_t> depth((size_t)R * nframes * npix);
    if (!read_all(argv[2], poses) || !read_all(argv[3], depth) || !read_all(argv[4], kinv)) return 3;
    uint16_t *d_depth = nullptr;
    if (hipMalloc((void **)&d_depth, depth.size() * sizeof(uint16_t)) != hipSuccess) return 4;
    if (hipMemcpy(d_depth, depth.data(), depth.size() * sizeof(uint16_t), hipMemcpyHostToDevice) != hipSuccess) return 4;
    HashTableParams p;
    vh_default_params(&p);
    p.numBuckets = (uint32_t)std::atoi(argv[9]);
    p.numVoxelBlocks = (uint32_t)std::atoi(argv[10]);
    const std::string prefix = argv[11];
    char id[VH_DIST_ID_BYTES];
    SDF_Hashtable::loopbackId(id);
    // the ranks are created one after the other (creation is not a collective on the loop-back transport) ...
    std::vector<std::unique_ptr<SDF_Hashtable>> ranks;
    for (int r = 0; r < R; ++r) ranks.emplace_back(new SDF_Hashtable(p, W, H, VH_SEM_PINHOLE, r, R, batch, id, kinv.data()));
    std::vector<int> status(R, 0);
    std::vector<float *> d_out(R, nullptr);
    for (int r = 0; r < R; ++r)
        if (hipMalloc((void **)&d_out[r], npix * sizeof(float)) != hipSuccess) return 4;
    // ... and driven by one thread each: every exchange and every raycast round is a collective of the R ranks
    std::vector<std::thread> threads;
    for (int r = 0; r < R; ++r)
        threads.emplace_back([&, r] {
            SDF_Hashtable &table = *ranks[r];
            std::vector<const uint16_t *> ptrs(batch);
            for (int s = 0; s < steps; ++s) {
                for (int b = 0; b < batch; ++b) ptrs[b] = d_depth + (((size_t)r * nframes) + (size_t)s * batch + b) * npix;
                table.integrateExchange(poses.data() + ((size_t)r * nframes + (size_t)s * batch) * 16, ptrs.data());
            }
            table.flush();
            float4x4 view(poses.data() + ((size_t)r * nframes + nframes - 1) * 16);      // the rank's last pose
            table.raycast(view, d_out[r]);                                              // through every shard
            table.flush();
            status[r] = 1;
        });
    for (auto &t : threads) t.join();
    int allocated = 0;
    unsigned overflow = 0;
    for (int r = 0; r < R; ++r) {
        if (!status[r]) return 5;
        const uint32_t per = (p.numBuckets + (uint32_t)R - 1) / (uint32_t)R;
        const uint32_t lo = (uint32_t)r * per, hi = std::min(p.numBuckets, lo + per);
        std::vector<VoxelEntry> entries((size_t)(hi - lo) * p.bucketSize);
        if (vh_download(ranks[r]->context(), VH_BUF_HASH_TABLE, entries.data(), entries.size() * sizeof(VoxelEntry)) != VH_OK) return 5;
        for (const VoxelEntry &e : entries) allocated += e.ptr != VH_FREE_BLOCK;
        vh_counters c;
        if (vh_get_counters(ranks[r]->context(), &c) != VH_OK) return 5;
        overflow += c.bin_overflow;
        std::vector<float> out(npix);
        if (hipMemcpy(out.data(), d_out[r], npix * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return 5;
        if (!write_all(prefix + "table" + std::to_string(r) + ".bin", entries) || !write_all(prefix + "depth" + std::to_string(r) + ".bin", out)) return 6;
    }
    std::printf("ranks=%d allocated=%d bin_overflow=%u\n", R, allocated, overflow);
    ranks.clear();
    (void)hipFree(d_depth);
    for (float *q : d_out) (void)hipFree(q);
    return 0;
}
