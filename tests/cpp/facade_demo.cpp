// Drives the C++ SDF_Hashtable facade the way Application.cpp:33-35,82-85 drives the
// reference class: default-constructed table, identity pose, integrate() of one vertex
// map (twice), then the raycast that stands in for SDFRenderer::render().
//   facade_demo <verts.bin: 640*480 float4>     prints "occupied=<n> allocated=<n> hits=<n>"
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "SDF_Hashtable.h"

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    const size_t n = 640 * 480;
    std::vector<vh_float4> h_verts(n);
    FILE *f = std::fopen(argv[1], "rb");
    if (!f || std::fread(h_verts.data(), sizeof(vh_float4), n, f) != n) return 3;
    std::fclose(f);
    vh_float4 *d_verts = nullptr;
    float *d_depth = nullptr;
    if (hipMalloc((void **)&d_verts, n * sizeof(vh_float4)) != hipSuccess) return 4;
    if (hipMalloc((void **)&d_depth, n * sizeof(float)) != hipSuccess) return 4;
    hipMemcpy(d_verts, h_verts.data(), n * sizeof(vh_float4), hipMemcpyHostToDevice);

    SDF_Hashtable table;                         // common.h defaults, REFERENCE semantics
    float4x4 pose;
    pose.setIdentity();                          // Application.cpp:82-83
    table.integrate(pose, d_verts, (const vh_float4 *)nullptr);
    table.integrate(pose, d_verts, (const vh_float4 *)nullptr);
    const int occupied = table.occupiedBlockCount();

    const HashTableParams &p = table.params();
    std::vector<VoxelEntry> entries((size_t)p.numBuckets * p.bucketSize);
    if (vh_download(table.context(), VH_BUF_HASH_TABLE, entries.data(), entries.size() * sizeof(VoxelEntry)) != VH_OK)
        return 5;
    int allocated = 0;
    for (const VoxelEntry &e : entries) allocated += e.ptr != VH_FREE_BLOCK;

    table.raycast(pose, d_depth);
    std::vector<float> depth(n);
    vh_synchronize(table.context());
    hipMemcpy(depth.data(), d_depth, n * sizeof(float), hipMemcpyDeviceToHost);
    int hits = 0;
    for (float z : depth) hits += z > 0.0f;
    // a second table through the batch / pipeline / silhouette members (SDF_Hashtable.h): two frames as one
    // batch, a third as a streaming pipelined frame left pending until flush(), then the block silhouettes
    int allocated2 = 0, covered = 0;
    {
        SDF_Hashtable t2;
        float poses[32];
        for (int k = 0; k < 2; ++k)
            for (int i = 0; i < 16; ++i) poses[16 * k + i] = (i % 5 == 0) ? 1.0f : 0.0f;
        const vh_float4 *ptrs[2] = {d_verts, d_verts};
        t2.integrateBatch(2, poses, ptrs, nullptr);
        t2.setOption("pipeline", 1);
        t2.integrate(pose, d_verts, (const vh_float4 *)nullptr);
        t2.flush();
        if (vh_download(t2.context(), VH_BUF_HASH_TABLE, entries.data(), entries.size() * sizeof(VoxelEntry)) != VH_OK)
            return 6;
        for (const VoxelEntry &e : entries) allocated2 += e.ptr != VH_FREE_BLOCK;
        float *d_back = nullptr;
        if (hipMalloc((void **)&d_back, n * sizeof(float)) != hipSuccess) return 4;
        t2.renderBlocks(pose, d_depth, d_back);
        vh_synchronize(t2.context());
        hipMemcpy(depth.data(), d_depth, n * sizeof(float), hipMemcpyDeviceToHost);
        for (float z : depth) covered += z > 0.0f;
        hipFree(d_back);
    }
    // a third table on the walk-free frame (option "flatten_variant" 4: the occupancy-index walk in place of flattenKernel's scan;
    // the same table and compact list by construction), three frames pipelined
    int allocated3 = 0, occupied3 = 0;
    {
        SDF_Hashtable t3;
        t3.setOption("flatten_variant", 4);
        float poses[48];
        for (int k = 0; k < 3; ++k)
            for (int i = 0; i < 16; ++i) poses[16 * k + i] = (i % 5 == 0) ? 1.0f : 0.0f;
        const vh_float4 *ptrs[3] = {d_verts, d_verts, d_verts};
        t3.integrateBatch(3, poses, ptrs, nullptr);
        if (vh_download(t3.context(), VH_BUF_HASH_TABLE, entries.data(), entries.size() * sizeof(VoxelEntry)) != VH_OK)
            return 7;
        for (const VoxelEntry &e : entries) allocated3 += e.ptr != VH_FREE_BLOCK;
        vh_counters c3;
        if (vh_get_counters(t3.context(), &c3) != VH_OK) return 8;
        occupied3 = c3.occupied;
    }
    std::printf("occupied=%d allocated=%d hits=%d allocated2=%d covered=%d allocated3=%d occupied3=%d\n", occupied, allocated, hits, allocated2,
                covered, allocated3, occupied3);
    hipFree(d_verts);
    hipFree(d_depth);
    return 0;
}
