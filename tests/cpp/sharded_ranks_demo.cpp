// ONE rank of a C++ host of the sharded table, one PROCESS per rank over RCCL (include/voxelhash_dist.h through
// SDF_Hashtable's multi-GPU constructor) -- what a node with R GPUs runs, R times.  Rank 0 draws the communicator's id and
// leaves it in a file, the other ranks wait for that file (any out-of-band channel would do); every rank then feeds `steps`
// exchanges of `batch` uint16 sensor frames of ITS camera, flushes, dumps its shard's hash table and renders its view
// through every shard.  No Python, no torch in this process.
//   sharded_ranks_demo rank world id_file poses.bin depth.bin kinv.bin W H batch steps numBuckets numVoxelBlocks out_prefix
// poses.bin / depth.bin hold all ranks' frames, rank-major: [world][steps*batch][16] floats / [world][steps*batch][H*W] uint16.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include "SDF_Hashtable.h"

template <class T>
static bool read_at(const char *path, std::vector<T> &v, size_t firstElement)
{
    FILE *f = std::fopen(path, "rb");
    if (!f) return false;
    const bool ok = std::fseek(f, (long)(firstElement * sizeof(T)), SEEK_SET) == 0 && std::fread(v.data(), sizeof(T), v.size(), f) == v.size();
    std::fclose(f);
    return ok;
}

int main(int argc, char **argv)
{
    if (argc < 14) return 2;
    const int rank = std::atoi(argv[1]), world = std::atoi(argv[2]);
    const std::string idFile = argv[3], prefix = argv[13];
    const int W = std::atoi(argv[7]), H = std::atoi(argv[8]), batch = std::atoi(argv[9]), steps = std::atoi(argv[10]);
    const size_t npix = (size_t)W * H, nframes = (size_t)batch * steps;
    std::vector<float> poses(nframes * 16), kinv(9);
    std::vector<uint16_t> depth(nframes * npix);
    if (!read_at(argv[4], poses, (size_t)rank * nframes * 16) || !read_at(argv[5], depth, (size_t)rank * nframes * npix) || !read_at(argv[6], kinv, 0)) return 3;

    char id[VH_DIST_ID_BYTES];
    if (rank == 0) {
        SDF_Hashtable::uniqueId(id);
        const std::string tmp = idFile + ".tmp";
        FILE *f = std::fopen(tmp.c_str(), "wb");
        if (!f || std::fwrite(id, 1, sizeof id, f) != sizeof id) return 7;
        std::fclose(f);
        if (std::rename(tmp.c_str(), idFile.c_str()) != 0) return 7;          // (complete when it appears)
    } else {
        FILE *f = nullptr;
        for (int tries = 0; tries < 1200 && !(f = std::fopen(idFile.c_str(), "rb")); ++tries)
            std::this_thread::sleep_for(std::chrono::milliseconds(50));
        if (!f || std::fread(id, 1, sizeof id, f) != sizeof id) return 7;
        std::fclose(f);
    }

    uint16_t *d_depth = nullptr;
    float *d_out = nullptr;
    if (hipMalloc((void **)&d_depth, depth.size() * sizeof(uint16_t)) != hipSuccess) return 4;
    if (hipMalloc((void **)&d_out, npix * sizeof(float)) != hipSuccess) return 4;
    if (hipMemcpy(d_depth, depth.data(), depth.size() * sizeof(uint16_t), hipMemcpyHostToDevice) != hipSuccess) return 4;

    HashTableParams p;
    vh_default_params(&p);
    p.numBuckets = (uint32_t)std::atoi(argv[11]);
    p.numVoxelBlocks = (uint32_t)std::atoi(argv[12]);
    SDF_Hashtable table(p, W, H, VH_SEM_PINHOLE, rank, world, batch, id, kinv.data());
    std::vector<const uint16_t *> ptrs(batch);
    for (int s = 0; s < steps; ++s) {
        for (int b = 0; b < batch; ++b) ptrs[b] = d_depth + ((size_t)s * batch + b) * npix;
        table.integrateExchange(poses.data() + (size_t)s * batch * 16, ptrs.data());
    }
    table.flush();
    vh_counters c;
    if (vh_get_counters(table.context(), &c) != VH_OK) return 5;
    // this rank's slice of the table: buckets [rank*per, min((rank+1)*per, numBuckets))
    const uint32_t per = (p.numBuckets + (uint32_t)world - 1u) / (uint32_t)world;
    const uint32_t lo = (uint32_t)rank * per, hi = lo + per < p.numBuckets ? lo + per : p.numBuckets;
    std::vector<VoxelEntry> entries((size_t)(hi - lo) * p.bucketSize);
    if (vh_download(table.context(), VH_BUF_HASH_TABLE, entries.data(), entries.size() * sizeof(VoxelEntry)) != VH_OK) return 5;
    int allocated = 0;
    for (const VoxelEntry &e : entries) allocated += e.ptr != VH_FREE_BLOCK;
    FILE *f = std::fopen((prefix + "table" + std::to_string(rank) + ".bin").c_str(), "wb");
    if (!f || std::fwrite(entries.data(), sizeof(VoxelEntry), entries.size(), f) != entries.size()) return 6;
    std::fclose(f);
    float4x4 view(poses.data() + (nframes - 1) * 16);
    table.raycast(view, d_out);                          // this rank's view through every shard (collective)
    table.flush();
    std::vector<float> out(npix);
    if (hipMemcpy(out.data(), d_out, npix * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return 4;
    f = std::fopen((prefix + "depth" + std::to_string(rank) + ".bin").c_str(), "wb");
    if (!f || std::fwrite(out.data(), sizeof(float), npix, f) != npix) return 6;
    std::fclose(f);
    std::printf("rank=%d allocated=%d occupied=%d epoch=%u bin_overflow=%u\n", rank, allocated, c.occupied, c.epoch, c.bin_overflow);
    (void)hipFree(d_depth);
    (void)hipFree(d_out);
    return 0;
}
