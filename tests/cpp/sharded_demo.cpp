// A C++ host of the sharded table (include/voxelhash_dist.h through SDF_Hashtable's multi-GPU constructor), one rank:
// the process creates the RCCL communicator itself, feeds `steps` exchanges of `batch` uint16 sensor frames, flushes,
// dumps the shard's hash table and renders one view through vh_dist_raycast.  No Python, no torch in this process.
//   sharded_demo poses.bin depth.bin kinv.bin W H batch steps numBuckets numVoxelBlocks table_out.bin depth_out.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
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

int main(int argc, char **argv)
{
    if (argc < 12) return 2;
    const int W = std::atoi(argv[4]), H = std::atoi(argv[5]), batch = std::atoi(argv[6]), steps = std::atoi(argv[7]);
    const size_t npix = (size_t)W * H, nframes = (size_t)batch * steps;
    std::vector<float> poses(nframes * 16), kinv(9);
    std::vector<uint16_t> depth(nframes * npix);
    if (!read_all(argv[1], poses) || !read_all(argv[2], depth) || !read_all(argv[3], kinv)) return 3;
    uint16_t *d_depth = nullptr;
    float *d_out = nullptr;
    if (hipMalloc((void **)&d_depth, depth.size() * sizeof(uint16_t)) != hipSuccess) return 4;
    if (hipMalloc((void **)&d_out, npix * sizeof(float)) != hipSuccess) return 4;
    hipMemcpy(d_depth, depth.data(), depth.size() * sizeof(uint16_t), hipMemcpyHostToDevice);

    HashTableParams p;
    vh_default_params(&p);
    p.numBuckets = (uint32_t)std::atoi(argv[8]);
    p.numVoxelBlocks = (uint32_t)std::atoi(argv[9]);
    char id[VH_DIST_ID_BYTES];
    SDF_Hashtable::uniqueId(id);                         // (with several ranks: drawn by rank 0, sent to the others)
    SDF_Hashtable table(p, W, H, VH_SEM_PINHOLE, /*rank*/ 0, /*world*/ 1, batch, id, kinv.data());
    std::vector<const uint16_t *> ptrs(batch);
    for (int s = 0; s < steps; ++s) {
        for (int b = 0; b < batch; ++b) ptrs[b] = d_depth + ((size_t)s * batch + b) * npix;
        table.integrateExchange(poses.data() + (size_t)s * batch * 16, ptrs.data());
    }
    table.flush();
    vh_counters c;
    if (vh_get_counters(table.context(), &c) != VH_OK) return 5;
    std::vector<VoxelEntry> entries((size_t)p.numBuckets * p.bucketSize);
    if (vh_download(table.context(), VH_BUF_HASH_TABLE, entries.data(), entries.size() * sizeof(VoxelEntry)) != VH_OK) return 5;
    int allocated = 0;
    for (const VoxelEntry &e : entries) allocated += e.ptr != VH_FREE_BLOCK;
    FILE *f = std::fopen(argv[10], "wb");
    if (!f || std::fwrite(entries.data(), sizeof(VoxelEntry), entries.size(), f) != entries.size()) return 6;
    std::fclose(f);
    float4x4 view(poses.data() + 5 * 16);
    table.raycast(view, d_out);                          // this rank's view through every shard
    table.flush();
    std::vector<float> out(npix);
    hipMemcpy(out.data(), d_out, npix * sizeof(float), hipMemcpyDeviceToHost);
    f = std::fopen(argv[11], "wb");
    if (!f || std::fwrite(out.data(), sizeof(float), npix, f) != npix) return 6;
    std::fclose(f);
    std::printf("allocated=%d occupied=%d epoch=%u bin_overflow=%u\n", allocated, c.occupied, c.epoch, c.bin_overflow);
    hipFree(d_depth);
    hipFree(d_out);
    return 0;
}
