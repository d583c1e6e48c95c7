// vh_kernels.hip -- the hand-written gfx950 kernels of the voxel-hashing TSDF path, one file
// per stage of SDF_Hashtable::integrate (SDF_Hashtable.cpp:11-40) and its neighbours:
//
//   vh_alloc.hip       allocBlocks: per-pixel block key, wave-level run dedup, bucket probe,
//                      epoch-stamped claim (allocBlocksKernel + the locking half of
//                      insertVoxelEntry, VoxelUtils.cu:606-705, 418-456); commit of the winners
//                      (VoxelUtils.cu:447-453, 328-334); key generation for the multi-GPU exchange
//   vh_walk.hip        flattenIntoBuffer: the walk over the VoxelEntry array, wave-ballot
//                      compaction of allocated in-frustum entries (flattenKernel, :719-749),
//                      and the measured alternatives (wide chunks, occupancy index, ...)
//   vh_integrate.hip   integrateDepthMap: one 8^3 block per workgroup pass, 16-byte-per-lane
//                      voxel read-modify-write (integrateDepthMapKernel, :790-842)
//   vh_frame.hip       the fused frame: {claim || walk} and {commit + integrate} in two launches
//   vh_shard.hip       the multi-camera frame on a bucket-range shard
//   vh_raycast.hip     per-pixel march through the hash (stand-in for SDFRenderer::render,
//                      SDFRenderer.cpp:210-255)
//   vh_view.hip        raycast over shards: export of the blocks a view can touch, view table import
//   vh_blocks.hip      block silhouettes: per-pixel nearest front / farthest back face of the allocated
//                      blocks' cubes (SDFRenderer::drawToFrontAndBack, SDFRenderer.cpp:165-208)
//   vh_gc.hip          block deletion / garbage collection (deleteVoxelEntry :544-604 done correctly)
//   vh_preprocess.hip  depth -> vertex / normal maps (preProcess, CameraTrackingUtils.cu:50-120),
//                      table set-up kernels (VoxelUtils.cu:151-166), device-side test hook
//   vh_icp.hip         frame-to-frame point-to-plane ICP: correspondences + Jacobian + J^T J / J^T r in one
//                      pass (CameraTrackingUtils.cu:131-185, Solver.cu:19-54, Solver.cpp:81-90)
//
// All of it is integer / fp32 scalar work bound by HBM traffic, latency or VALU issue; there is
// no contraction to hand to MFMA.  Built with -ffp-contract=off (see vh_device.h).
#include "vh_device.h"

#include "vh_alloc.hip"
#include "vh_walk.hip"
#include "vh_integrate.hip"
#include "vh_frame.hip"
#include "vh_shard.hip"
#include "vh_raycast.hip"
#include "vh_raycast_coop.hip"
#include "vh_view.hip"
#include "vh_gc.hip"
#include "vh_blocks.hip"
#include "vh_preprocess.hip"
#include "vh_icp.hip"
