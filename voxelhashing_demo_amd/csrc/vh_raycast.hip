// vh_raycast.hip -- raycast through the hash (stand-in for SDFRenderer::render).
// Part of libvoxelhash_hip.so (gfx950); included by vh_kernels.hip after vh_device.h.
#pragma once

namespace vh {

// ---------------------------------------------------------------------------
// raycast
// ---------------------------------------------------------------------------
// getVoxelEntry4Block, live half (VoxelUtils.cu:362-382)
__device__ __forceinline__ int lookup_block(const FrameParams &fp, const DevPtrs &dp, int kx, int ky, int kz)
{
    const uint32_t h = hash_block(kx, ky, kz, fp.numBuckets);
    if (h < fp.bucketLo || h >= fp.bucketHi) return VH_FREE_BLOCK;
    const uint32_t local = h - fp.bucketLo;
    // One bit per bucket ("holds at least one entry", set by the commit phase): a few hundred
    // KB that stay in L2, while the table itself is >100 MB.  Nearly every block a ray crosses
    // is empty space and is answered here without touching the table.
    if (!((dp.bucketBits[local >> 5] >> (local & 31u)) & 1u)) return VH_FREE_BLOCK;
    const VoxelEntry *bucket = dp.table + (size_t)local * fp.bucketSize;
    for (uint32_t i = 0; i < fp.bucketSize; ++i) {
        const VoxelEntry e = bucket[i];
        if (e.ptr == VH_FREE_BLOCK) return VH_FREE_BLOCK;      // prefix property
        if (e.pos[0] == kx && e.pos[1] == ky && e.pos[2] == kz) return e.ptr;
    }
    return VH_FREE_BLOCK;
}

// Spec (DESIGN.md "raycast", oracle/vh_oracle.c vho_raycast): samples at camera
// depth t_i = tMin + i*voxelSize, nearest-voxel classification, first pair of
// consecutive valid samples with sdf_prev > 0 >= sdf_cur, linear interpolation.
// 16x16-pixel tiles: a wave is a 16x4 patch of neighbouring rays, which walk
// the same blocks and keep the bucket / voxel lines hot in L2.
constexpr float kSkipMargin = 0.01f;     // voxels; see the empty-block skip below
// kRayBatch (template): in-block samples whose voxels are fetched together

template <int kRayBatch, bool kFastDiv>
__global__ __launch_bounds__(256) void raycast_kernel(const FrameParams fp, const DevPtrs dp, float fx, float fy,
                                                      float cx, float cy, float tMin, int nSteps,
                                                      float *__restrict__ depthOut)
{
    const int u = blockIdx.x * 16 + (threadIdx.x & 15);
    const int v = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (u >= fp.width || v >= fp.height) return;
    const float dx = ((float)u - cx) / fx;
    const float dy = ((float)v - cy) / fy;
    const float dt = fp.voxelSize;
    const float invDt = __builtin_amdgcn_rcpf(dt) * (1.0f - 1.0e-6f);   // never over-estimates a step count
    const float rcpVoxel = 1.0f / fp.voxelSize;                          // correctly rounded (world2voxel1_fast)
    // world-space ray per unit of camera depth (only used to bound empty-block skips)
    const float dirX = fp.T[0] * dx + fp.T[1] * dy + fp.T[2];
    const float dirY = fp.T[4] * dx + fp.T[5] * dy + fp.T[6];
    const float dirZ = fp.T[8] * dx + fp.T[9] * dy + fp.T[10];
    const float rayD[3] = {dirX, dirY, dirZ}, rayO[3] = {fp.T[3], fp.T[7], fp.T[11]};
    float invD[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) invD[a] = (rayD[a] != 0.0f) ? __builtin_amdgcn_rcpf(rayD[a]) : 0.0f;
    bool prevValid = false, haveKey = false, found = false, haveMacro = false, macroEmpty = false;
    float prevSdf = 0.0f, prevT = 0.0f, hit = 0.0f;
    int ckx = 0, cky = 0, ckz = 0, cptr = VH_FREE_BLOCK;
    int cmx = 0, cmy = 0, cmz = 0;
    for (int i = 0; i < nSteps; ++i) {
        const float tt = tMin + (float)i * dt;
        const float4 pw = mat4_mul(fp.T, dx * tt, dy * tt, tt, 1.0f);
        const int vx = kFastDiv ? world2voxel1_fast(pw.x, fp.voxelSize, rcpVoxel) : world2voxel1(pw.x, fp.voxelSize);
        const int vy = kFastDiv ? world2voxel1_fast(pw.y, fp.voxelSize, rcpVoxel) : world2voxel1(pw.y, fp.voxelSize);
        const int vz = kFastDiv ? world2voxel1_fast(pw.z, fp.voxelSize, rcpVoxel) : world2voxel1(pw.z, fp.voxelSize);
        const int kx = voxel2block1(vx), ky = voxel2block1(vy), kz = voxel2block1(vz);
        if (!haveKey || kx != ckx || ky != cky || kz != ckz) {
            ckx = kx; cky = ky; ckz = kz;
            haveKey = true;
            const int mx = kx >> 2, my = ky >> 2, mz = kz >> 2;          // macro cell of 4x4x4 blocks
            if (!haveMacro || mx != cmx || my != cmy || mz != cmz) {
                cmx = mx; cmy = my; cmz = mz;
                haveMacro = true;
                const uint32_t hm = macro_hash(mx, my, mz);
                macroEmpty = !((dp.macroBits[hm >> 5] >> (hm & 31u)) & 1u);
            }
            cptr = macroEmpty ? VH_FREE_BLOCK : lookup_block(fp, dp, kx, ky, kz);
        }
        if (cptr == VH_FREE_BLOCK) {
            // Empty block: every further sample inside it is invalid too, so jump to the last
            // sample that is CERTAINLY still inside (cell shrunk by kSkipMargin voxels per side:
            // 1e-2 voxel = 2e-4 m at 2 cm voxels, against ~1e-6 m of fp32 difference between this
            // linear ray model and the sample positions above).  Skipping only such samples
            // leaves the result unchanged.
            prevValid = false;
            float tExit = 3.0e38f;
            // an empty macro cell (no block in 4x4x4) is skipped whole: 32 voxels per side
            const int cell[3] = {macroEmpty ? cmx * 32 : kx * 8, macroEmpty ? cmy * 32 : ky * 8,
                                 macroEmpty ? cmz * 32 : kz * 8};
            const float span = macroEmpty ? 31.5f : 7.5f;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                // the cell spans voxel centres c .. c+span-0.5, i.e. world [(c-0.5)vs, (c+span)vs)
                const float lo = ((float)cell[a] - 0.5f + kSkipMargin) * fp.voxelSize;
                const float hi = ((float)cell[a] + span - kSkipMargin) * fp.voxelSize;
                // approximate reciprocals (1 ulp) are fine here: the margin absorbs them
                if (rayD[a] > 0.0f) tExit = __builtin_fminf(tExit, (hi - rayO[a]) * invD[a]);
                else if (rayD[a] < 0.0f) tExit = __builtin_fminf(tExit, (lo - rayO[a]) * invD[a]);
            }
            const float steps = (tExit - tMin) * invDt;     // last sample index at or before tExit
            if (steps > (float)i && steps < 2.0e9f) i = min((int)steps, nSteps - 1);
            continue;
        }
        // Present block: the voxel of sample i and of the next kRayBatch-1 samples that still
        // fall into this block are fetched together (their addresses do not depend on each
        // other, only the hit test is sequential), so a ray pays one memory latency per batch
        // instead of one per sample.  Samples are then classified strictly in order.
        float bt[kRayBatch];
        Voxel bs[kRayBatch];
        bool inBlock[kRayBatch];
#pragma unroll
        for (int j = 0; j < kRayBatch; ++j) {
            bt[j] = tMin + (float)(i + j) * dt;
            const float4 pj = mat4_mul(fp.T, dx * bt[j], dy * bt[j], bt[j], 1.0f);
            const int jx = kFastDiv ? world2voxel1_fast(pj.x, fp.voxelSize, rcpVoxel) : world2voxel1(pj.x, fp.voxelSize);
            const int jy = kFastDiv ? world2voxel1_fast(pj.y, fp.voxelSize, rcpVoxel) : world2voxel1(pj.y, fp.voxelSize);
            const int jz = kFastDiv ? world2voxel1_fast(pj.z, fp.voxelSize, rcpVoxel) : world2voxel1(pj.z, fp.voxelSize);
            inBlock[j] = (i + j < nSteps) && voxel2block1(jx) == kx && voxel2block1(jy) == ky &&
                         voxel2block1(jz) == kz;
            const int lx = (int)((uint32_t)jx - (uint32_t)kx * 8u);
            const int ly = (int)((uint32_t)jy - (uint32_t)ky * 8u);
            const int lz = (int)((uint32_t)jz - (uint32_t)kz * 8u);
            bs[j] = inBlock[j] ? dp.blocks[(size_t)cptr + (size_t)(lz * 64 + ly * 8 + lx)] : Voxel{0.0f, 0.0f};
        }
        bool done = false;
        int used = 0;
#pragma unroll
        for (int j = 0; j < kRayBatch; ++j) {
            if (done || !inBlock[j]) { done = true; continue; }   // first sample outside: back to the general path
            used = j + 1;
            if (!(bs[j].weight > 0.0f)) { prevValid = false; continue; }
            if (prevValid && prevSdf > 0.0f && bs[j].sdf <= 0.0f) {
                hit = prevT + (dt * prevSdf) / (prevSdf - bs[j].sdf);
                found = true;
                done = true;
                continue;
            }
            prevValid = true; prevSdf = bs[j].sdf; prevT = bt[j];
        }
        if (found) break;
        i += used - 1;            // sample i itself is always in the block, so used >= 1
    }
    depthOut[(size_t)v * fp.width + u] = hit;
}

}  // namespace vh
