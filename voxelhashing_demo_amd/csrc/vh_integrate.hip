// vh_integrate.hip -- integrateDepthMap: 8^3-block TSDF update.
// Part of libvoxelhash_hip.so (gfx950); included by vh_kernels.hip after vh_device.h.
#pragma once

namespace vh {

// ---------------------------------------------------------------------------
// integrateDepthMap
// ---------------------------------------------------------------------------
// A workgroup of 256 lanes owns one 8^3 block per pass: lane t updates voxels
// 2t and 2t+1 (neighbours in x), so the block moves as 16-byte-per-lane
// coalesced loads and stores (4 KiB in, 4 KiB out) instead of the reference's
// 8-byte accesses.  The occupied count never leaves the device: the grid is a
// fixed size and strides over the compact list.
// Where the camera z of a pixel comes from:
//   DepthPlane  depth(x,y) = base[stride*(y*W+x)]: stride 4 from &verts[0].z (float4 vertex map),
//               stride 1 for the camera-z plane of a float camera packet
//   DepthSensor the uint16 sensor image of a VH_PACKET_U16 packet: z = (K_inv row 2 . (x,y,1)) * (d / unit),
//               the operations of calculateVertexPositions (CameraTrackingUtils.cu:63-73) in their order
struct DepthPlane {
    const float *__restrict__ base;
    int stride;
    __device__ __forceinline__ float at(int sx, int sy, int width) const
    {
        return base[(size_t)stride * ((size_t)sy * width + sx)];
    }
};

struct DepthSensor {
    const uint16_t *__restrict__ image;
    float k6, k7, k8, unit;
    __device__ __forceinline__ float at(int sx, int sy, int width) const
    {
        const float d = (float)image[(size_t)sy * width + sx] / unit;
        const float pz = k6 * (float)sx + k7 * (float)sy + k8 * 1.0f;
        return pz * d;
    }
};

template <class Depth>
__device__ __forceinline__ bool tsdf_update(const FrameParams &fp, const float *Tinv, const Depth &src, int vx, int vy,
                                            int vz, float &sdfOut, float &wOut)
{
    float cx, cy, cz;
    if (fp.semantics == VH_SEM_REFERENCE) {
        // VoxelUtils.cu:797-800: inverse pose on the voxel INDEX, truncate, then metres
        const float4 r = mat4_mul(Tinv, (float)vx, (float)vy, (float)vz, 1.0f);
        cx = (float)f2i_rz(r.x) * fp.voxelSize;
        cy = (float)f2i_rz(r.y) * fp.voxelSize;
        cz = (float)f2i_rz(r.z) * fp.voxelSize;
    } else {
        const float4 r = mat4_mul(Tinv, (float)vx * fp.voxelSize, (float)vy * fp.voxelSize,
                                  (float)vz * fp.voxelSize, 1.0f);
        cx = r.x; cy = r.y; cz = r.z;
    }
    int sx, sy;
    project(fp.proj, cx, cy, cz, sx, sy);                                        // :801
    if (sx < 0 || sx >= fp.width || sy < 0 || sy >= fp.height) return false;     // :803
    const float depth = src.at(sx, sy, fp.width);                                 // :805
    if (depth <= 0.0f) return false;                                             // :806
    float sdf = depth - cz;                                                      // :813
    float trunc = fp.truncation;                                                 // :815
    if (fp.flags & kFlagDepthTruncation) trunc = fp.truncation + (fp.truncScale * depth);   // its commented half = getTruncation, :261-264
    if (!(sdf > -trunc)) return false;                                           // :818
    sdf = (sdf >= 0.0f) ? __builtin_fminf(trunc, sdf) : __builtin_fmaxf(-trunc, sdf);
    float cw = 0.1f;                                                             // :829
    if (fp.flags & kFlagWeightSample) {                                          // the commented :827 with :808-811 (double arithmetic there)
        const float zeroOne = (depth - 0.5f) / (5.0f - 0.5f);
        cw = __builtin_fmaxf((float)((double)fp.weightSample * 1.5 * (1.0 - (double)zeroOne)), 1.0f);
    }
    // combineVoxel, :779-787
    const float ow = wOut, os = sdfOut;
    sdfOut = ((os * ow) + (sdf * cw)) / (ow + cw);
    wOut = __builtin_fminf(fp.weightMax, ow + cw);
    return true;
}

// the 256 lanes of a workgroup update the 8^3 block of entry e; the depth comes from the float4
// vertex map (DepthPlane on &verts[0].z) or straight from the sensor image (DepthSensor)
template <class Depth>
__device__ __forceinline__ void integrate_block(const FrameParams &fp, const DevPtrs &dp, const VoxelEntry &e,
                                                const Depth &src)
{
    const int lin = 2 * (int)threadIdx.x;        // linearizeVoxelPos: z*64 + y*8 + x  (:311-317)
    const int tx = lin & 7, ty = (lin >> 3) & 7, tz = lin >> 6;
    const int bx = (int)((uint32_t)e.pos[0] * 8u) + tx;     // block2Voxel + threadIdx (:793-796)
    const int by = (int)((uint32_t)e.pos[1] * 8u) + ty;
    const int bz = (int)((uint32_t)e.pos[2] * 8u) + tz;
    float4 *cell = reinterpret_cast<float4 *>(dp.blocks + (size_t)e.ptr + lin);
    float4 v = *cell;                            // {sdf0, w0, sdf1, w1}
    const bool u0 = tsdf_update(fp, fp.Tinv, src, bx, by, bz, v.x, v.y);
    const bool u1 = tsdf_update(fp, fp.Tinv, src, bx + 1, by, bz, v.z, v.w);
    if (u0 || u1) *cell = v;
}

// The blocks list[first], list[first + stride], ... < count, one per workgroup pass.
// Measured and dropped in round 2 (in-process A/B on C3, 4 261 visible blocks, launch 2 of the
// two-launch frame): one block per WAVE with four 16-byte loads per lane (17.8 us against 14.4: eight
// voxels per lane in a row make the dependent chain longer than the extra loads in flight shorten it);
// software pipelining over the list, voxels of block k+1 and entry of block k+2 in flight while block k
// is updated (14.2 us against 13.6 without it: the workgroups of a 2048-4096 grid already overlap each
// other's chains); staging the 4 KiB through LDS as a prefetch buffer (global_load_lds_dwordx4, block k+1 into a double
// buffer while block k is updated; rounds 3 and 5) was slower in every frame form -- C3 launch 2 13.4 vs 11.6 us, pipelined C3
// 71.4 vs 70.3, loaded C2 26.2 vs 25.6, walk-free C3 29.4 vs 27.3: every voxel is read once and written once by the same
// lane, so there is no reuse for LDS to serve (north_star's "one 8^3 block staged into LDS per workgroup" is therefore an
// accepted, measured deviation: README.md; the code is in git history, DESIGN_LOG.md names the commit).
template <class Depth>
// countB > 0: the list has two ends (CompactOut, vh_walk.hip): entries 0 .. count-1 from the front, countB more
// from the back of the numEntries-entry buffer
__device__ __forceinline__ void integrate_list(const FrameParams &fp, const DevPtrs &dp, const VoxelEntry *__restrict__ list,
                                               int count, int first, int stride, const Depth &src, int countB = 0,
                                               uint32_t numEntries = 0)
{
    for (int k = first; k < count + countB; k += stride)
        integrate_block(fp, dp, k < count ? list[k] : list[numEntries - 1u - (uint32_t)(k - count)], src);
}

// The dense list of the boundary: end B of a two-ended list moved behind end A and what the commit phase appended
// to it.  The order inside the list is free (the reference's is an atomic race, VoxelUtils.cu:737-746), so only the
// entries of end B that lie beyond the dense range [0, total) move, into the gap [base, ...) below end B: source and
// destination ranges never overlap, however full the table is (end B = [n - nB, n), gap = [base, n - nB) when the
// two ends have met past `total`).  dst != dp.compact (the frame's list sits in the second buffer of the pipelined
// frames, the boundary's pointer names the first): end A is copied too and all of end B moves.
// counterB < 0: the list is dense already (step-level flatten); only the copy to dst remains.
__global__ __launch_bounds__(256) void compact_fold_kernel(const DevPtrs dp, VoxelEntry *dst, uint32_t numEntries,
                                                           int counterA, int counterB, int counterNew)
{
    const uint32_t nB = counterB >= 0 ? (uint32_t)dp.counters[counterB] : 0u;
    const uint32_t base = (uint32_t)dp.counters[counterA] + (counterNew >= 0 ? (uint32_t)dp.counters[counterNew] : 0u);
    const uint32_t total = base + nB;
    const uint32_t t = blockIdx.x * 256u + threadIdx.x, stride = gridDim.x * 256u;
    if (dst != dp.compact) {
        for (uint32_t k = t; k < base; k += stride) dst[k] = dp.compact[k];
        for (uint32_t k = t; k < nB; k += stride) dst[base + k] = dp.compact[numEntries - nB + k];
        return;
    }
    const uint32_t lo = max(total, numEntries - nB);          // entries of end B at or beyond `total`: [lo, n)
    for (uint32_t k = t; k < numEntries - lo; k += stride) dst[base + k] = dp.compact[lo + k];
}

// DepthPlane over the .z of a float4 vertex map
__host__ __device__ __forceinline__ DepthPlane vertex_depth(const float4 *__restrict__ verts)
{
    return DepthPlane{reinterpret_cast<const float *>(verts) + 2, 4};   // &verts[0].z
}

template <class Depth>
__global__ __launch_bounds__(256) void integrate_kernel(const FrameParams fp, const DevPtrs dp, const Depth verts)
{
    integrate_list(fp, dp, dp.compact, dp.counters[kCompactCount], (int)blockIdx.x, (int)gridDim.x, verts);
}

}  // namespace vh
