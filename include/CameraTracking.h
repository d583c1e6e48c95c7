/*
 * CameraTracking.h -- C++ host facade with the reference's tracker interface
 * (CameraTracking.h:36-59) on top of the C-ABI in voxelhash.h:
 *
 *     CameraTracking tracker(640, 480);
 *     SetCameraIntrinsic(K, K_inv);                          // CameraTracking.cpp:16
 *     tracker.Align(d_input, d_inputNormals, d_target, d_targetNormals, d_depthInput, d_depthTarget);
 *     float4x4 deltaT = tracker.getTransform();              // Application.cpp:76
 *
 * The reference returns an Eigen column-major Matrix4x4f; here it is the row-major float4x4 that
 * SDF_Hashtable::integrate takes.  One fused GPU pass per round replaces FindCorrespondences +
 * CalculateJacAndResKernel + cublasSgemv + cublasSsyrk; thresholds and the round limit are the
 * reference's (common.h:12, CameraTracking.h:40).
 */
#ifndef CAMERA_TRACKING_H
#define CAMERA_TRACKING_H

#include <cstdint>

#include "SDF_Hashtable.h"

class CameraTracking {
    vh_icp *icp_;
    int width, height;
    int maxIters = 20;                 /* CameraTracking.h:40 */
    float4x4 deltaTransform;
    float K_[9];
    float distThres_ = 0.08f;          /* common.h:12 */
    int flags_ = 0;
    float globalCorrespondenceError = 0.0f;

public:
    CameraTracking(int w, int h);
    ~CameraTracking();
    CameraTracking(const CameraTracking &) = delete;
    CameraTracking &operator=(const CameraTracking &) = delete;

    /* intrinsics for the projective pairing, row-major 3x3 (the reference reads the __constant__ K
     * set by SetCameraIntrinsic; call this with the same matrix) */
    void setIntrinsic(const float K[9]);
    /* VH_ICP_ABS_DISTANCE | VH_ICP_NEED_TARGET; 0 = the reference's pairing rules */
    void setFlags(int flags) { flags_ = flags; }
    void setStream(void *hipStream);

    /* CameraTracking.cpp:27-69.  The depth images and the input normals are unused there as well. */
    void Align(vh_float4 *d_input, vh_float4 *d_inputNormals, vh_float4 *d_target, vh_float4 *d_targetNormals,
               const uint16_t *d_depthInput, const uint16_t *d_depthTarget);
    float4x4 getTransform() { return deltaTransform; }
    float lastError() const { return globalCorrespondenceError; }
};

#endif
