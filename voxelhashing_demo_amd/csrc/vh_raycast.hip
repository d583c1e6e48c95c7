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
    if (fp.flags & kFlagOverflow) {
        // (a bucket with a chain has an allocated last slot, so its bit is set; bits are per slot residence)
        uint32_t prev;
        const uint32_t at = find_entry_overflow(fp, dp.table, owned_entries(fp), local, kx, ky, kz, prev);
        return at == ~0u ? VH_FREE_BLOCK : dp.table[at].ptr;
    }
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
//
// Empty space.  A sample in an absent block (or in a macro cell of 4x4x4 blocks that holds no
// block) is invalid whatever its voxel index, so whole runs of such samples are skipped -- but
// only samples that are CERTAINLY in the empty cell: the linear ray model o + d*t used for the
// bounds differs from the sample positions T*(dx*t, dy*t, t) by ~1e-6 m, and the cell is shrunk by
// kSkipMargin voxels per side (1e-2 voxel = 2e-4 m at 2 cm voxels) when it is intersected.  The
// ray jumps to the last sample certainly inside; the next sample is evaluated exactly, like the
// oracle evaluates every sample, and tells which cell comes next.
// Measured and dropped: adopting the neighbouring cell after a clear single-face
// exit without evaluating that sample (bit-equal, but the lanes of a wave then sit in different
// code paths: 47 -> 64 us), fetching a ray's next in-block voxels together, a divide-free voxel index, and the exact
// division by voxelSize through a hoisted refined reciprocal (div_fixed: 47.3 vs 47.3 us).  Half-filled waves
// (32 rays per wave, twice the waves) take 69.7 instead of 47.0 us: the kernel is bound by instruction issue
// per SIMD about as much as by its load chains, so more waves for the same rays do not pay.  Rotating the loop by one
// sample (a voxel read issued in iteration i and judged in iteration i+1, behind the next sample's arithmetic;
// bit-equal) takes 51.0 instead of 47.2 us.
constexpr float kSkipMargin = 0.01f;     // voxels

// kPatch: pixels of a wave inside the 16x16 tile: 0 = 16x4 rows, 1 = 8x8 square
template <int kPatch>
__global__ __launch_bounds__(256) void raycast_kernel(const FrameParams fp, const DevPtrs dp, float fx, float fy,
                                                      float cx, float cy, float tMin, int nSteps,
                                                      float *__restrict__ depthOut, int xcdAware)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // Workgroups are handed to the 8 XCDs round robin (workgroup b runs on XCD b % 8, each with its own
    // L2).  With xcdAware the tiles are renumbered so that an XCD gets a contiguous run of image tiles:
    // neighbouring rays, which sample the same blocks, then share an L2.
    int tx = blockIdx.x, ty = blockIdx.y;
    if (xcdAware) {
        const int n = gridDim.x * gridDim.y, b = blockIdx.y * gridDim.x + blockIdx.x;
        if ((n & 7) == 0) {
            const int r = (b & 7) * (n >> 3) + (b >> 3);
            ty = r / (int)gridDim.x;
            tx = r - ty * (int)gridDim.x;
        }
    }
    const int u = tx * 16 + (kPatch == 0 ? (int)(threadIdx.x & 15) : (wave & 1) * 8 + (lane & 7));
    const int v = ty * 16 + (kPatch == 0 ? (int)(threadIdx.x >> 4) : (wave >> 1) * 8 + (lane >> 3));
    if (u >= fp.width || v >= fp.height) return;
    const float dx = ((float)u - cx) / fx;
    const float dy = ((float)v - cy) / fy;
    const float dt = fp.voxelSize;
    const float invDt = __builtin_amdgcn_rcpf(dt) * (1.0f - 1.0e-6f);   // never over-estimates a step count
    // world-space ray per unit of camera depth (only used to bound empty-cell skips)
    const float rayD[3] = {fp.T[0] * dx + fp.T[1] * dy + fp.T[2], fp.T[4] * dx + fp.T[5] * dy + fp.T[6],
                           fp.T[8] * dx + fp.T[9] * dy + fp.T[10]};
    const float rayO[3] = {fp.T[3], fp.T[7], fp.T[11]};
    float invD[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) invD[a] = (rayD[a] != 0.0f) ? __builtin_amdgcn_rcpf(rayD[a]) : 0.0f;
    bool prevValid = false, haveKey = false, haveMacro = false, macroEmpty = false;
    float prevSdf = 0.0f, prevT = 0.0f, hit = 0.0f;
    int ckx = 0, cky = 0, ckz = 0, cptr = VH_FREE_BLOCK;
    int cmx = 0, cmy = 0, cmz = 0;
    for (int i = 0; i < nSteps; ++i) {
        const float tt = tMin + (float)i * dt;
        const float4 pw = mat4_mul(fp.T, dx * tt, dy * tt, tt, 1.0f);
        const int vx = world2voxel1(pw.x, fp.voxelSize);
        const int vy = world2voxel1(pw.y, fp.voxelSize);
        const int vz = world2voxel1(pw.z, fp.voxelSize);
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
        // present block: classify the sample from its voxel
        const int lx = (int)((uint32_t)vx - (uint32_t)kx * 8u);
        const int ly = (int)((uint32_t)vy - (uint32_t)ky * 8u);
        const int lz = (int)((uint32_t)vz - (uint32_t)kz * 8u);
        const Voxel s = dp.blocks[(size_t)cptr + (size_t)(lz * 64 + ly * 8 + lx)];
        if (!(s.weight > 0.0f)) { prevValid = false; continue; }
        if (prevValid && prevSdf > 0.0f && s.sdf <= 0.0f) {
            hit = prevT + (dt * prevSdf) / (prevSdf - s.sdf);
            break;
        }
        prevValid = true; prevSdf = s.sdf; prevT = tt;
    }
    depthOut[(size_t)v * fp.width + u] = hit;
}

}  // namespace vh
