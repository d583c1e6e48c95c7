"""Host-side mirror of the reference's CameraTracking (CameraTracking.h:36-59) for Python callers:
frame-to-frame point-to-plane ICP on the GPU through the C-ABI (vh_icp_*), plus the helpers that
turn a raycast of the model into an ICP target (frame-to-model tracking).  No CPU fallback."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L

ICP_ABS_DISTANCE, ICP_NEED_TARGET = 1, 2
DIST_THRES = 0.08          # common.h:12
MAX_ITERS = 20             # CameraTracking.h:40


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def se3_exp(twist) -> np.ndarray:
    t, T = np.ascontiguousarray(twist, np.float64), np.zeros(16)
    L.load().vh_se3_exp(_dp(t), _dp(T))
    return T.reshape(4, 4)


def se3_log(T) -> np.ndarray:
    T, t = np.ascontiguousarray(np.asarray(T, np.float64).reshape(16)), np.zeros(6)
    L.load().vh_se3_log(_dp(T), _dp(t))
    return t


def system_arrays(sys: L.IcpSystem):
    return (np.array(sys.JTJ, np.float64).reshape(6, 6), np.array(sys.JTr, np.float64), float(sys.error),
            int(sys.count))


def icp_solve(JTJ, JTr, estimate):
    """-> (ok, new estimate): update = -(JTJ^-1 JTr), estimate = log(exp(update) exp(estimate))."""
    sys = L.IcpSystem()
    sys.JTJ[:] = np.asarray(JTJ, np.float64).reshape(36).tolist()
    sys.JTr[:] = np.asarray(JTr, np.float64).reshape(6).tolist()
    est = np.ascontiguousarray(estimate, np.float64).copy()
    rc = L.load().vh_icp_solve(C.byref(sys), _dp(est))
    return rc == 0, est


def depth_to_maps(depth, k_inv, positions, normals, stream=None):
    """float depth [H, W] in metres on the device -> float4 vertex and normal maps (in place)."""
    H, W = depth.shape
    k = np.ascontiguousarray(np.asarray(k_inv, np.float32).reshape(9))
    handle = 0 if stream is None else (stream if isinstance(stream, int) else stream.cuda_stream)
    L.check(L.load().vh_depth_to_maps(_ptr(depth), _fp(k), W, H, _ptr(positions), _ptr(normals), C.c_void_p(handle)),
            "vh_depth_to_maps")
    return positions, normals


class CameraTracking:
    """CameraTracking(width, height); Align(input, target, target_normals) -> getTransform()."""

    def __init__(self, width: int, height: int, K, device: int = -1, stream=None, dist_thres: float = DIST_THRES,
                 max_iters: int = MAX_ITERS, flags: int = 0):
        self._lib = L.load()
        self.width, self.height = width, height
        self.K = np.ascontiguousarray(np.asarray(K, np.float32).reshape(9))
        self.dist_thres, self.max_iters, self.flags = dist_thres, max_iters, flags
        h = C.c_void_p()
        L.check(self._lib.vh_icp_create(width, height, device, C.byref(h)), "vh_icp_create")
        self._h = h
        if stream is not None:
            L.check(self._lib.vh_icp_set_stream(self._h, C.c_void_p(stream if isinstance(stream, int)
                                                                      else stream.cuda_stream)), "vh_icp_set_stream")
        self.delta = np.eye(4, dtype=np.float32)
        self.last = None
        self.iterations = 0

    def close(self):
        if self._h:
            self._lib.vh_icp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def build_system(self, inp, target, target_normals, delta):
        d = np.ascontiguousarray(np.asarray(delta, np.float32).reshape(16))
        sys = L.IcpSystem()
        L.check(self._lib.vh_icp_build_system(self._h, _ptr(inp), _ptr(target), _ptr(target_normals), _fp(d),
                                              _fp(self.K), self.dist_thres, self.flags, C.byref(sys)),
                "vh_icp_build_system")
        return system_arrays(sys)

    def correspondences(self, inp, target, target_normals, delta, corres, corres_normals, residuals):
        d = np.ascontiguousarray(np.asarray(delta, np.float32).reshape(16))
        sys = L.IcpSystem()
        L.check(self._lib.vh_icp_correspondences(self._h, _ptr(inp), _ptr(target), _ptr(target_normals), _fp(d),
                                                 _fp(self.K), self.dist_thres, self.flags, _ptr(corres),
                                                 _ptr(corres_normals), _ptr(residuals), C.byref(sys)),
                "vh_icp_correspondences")
        return system_arrays(sys)

    def Align(self, inp, target, target_normals, start=None):
        """CameraTracking::Align: up to max_iters rounds; the result is kept in `delta` (getTransform)."""
        d = np.ascontiguousarray(np.asarray(np.eye(4) if start is None else start, np.float32).reshape(16)).copy()
        sys, it = L.IcpSystem(), C.c_int32()
        L.check(self._lib.vh_icp_align(self._h, _ptr(inp), _ptr(target), _ptr(target_normals), _fp(self.K),
                                       self.dist_thres, self.max_iters, self.flags, _fp(d), C.byref(sys),
                                       C.byref(it)), "vh_icp_align")
        self.delta, self.last, self.iterations = d.reshape(4, 4), system_arrays(sys), int(it.value)
        return self.delta

    def getTransform(self) -> np.ndarray:
        return self.delta


class FusionLoop:
    """The demo's frame order as a closed loop (Application.cpp:73-90: preProcess -> Align -> integrate, then the
    renderer reads the model; CameraTracking.cpp:26-69): per uint16 sensor frame

        vh_preprocess(depth)            -> input vertex map             (CameraTrackingUtils.cu:115-120)
        vh_icp_align(input, model maps) -> delta, pose = pose . delta   (frame-to-model: the target is the raycast)
        vh_integrate_depth(pose, depth) -> the model takes the frame    (SDF_Hashtable.cpp:11-40)
        vh_raycast_maps(pose)           -> depth + vertex + normal maps of the model for the next frame's Align

    all queued on ONE stream (the table's; the tracker is bound to it).  The only host synchronisation of a frame is the
    one inside vh_icp_align, which hands the 4x4 delta to the host -- the pose is a host value in the reference's
    interface (integrate(const float4x4&, ...)), so the next frame's calls cannot be queued before it is known.
    The first frame is integrated at `start_pose`."""

    def __init__(self, table, K, k_inv, stream=None, flags: int = ICP_ABS_DISTANCE | ICP_NEED_TARGET, max_iters: int = MAX_ITERS):
        import torch
        self.table, self.W, self.H = table, table.width, table.height
        self.k_inv = np.ascontiguousarray(np.asarray(k_inv, np.float32).reshape(9))
        self.trk = CameraTracking(self.W, self.H, K, stream=stream, flags=flags, max_iters=max_iters)
        dev = torch.device("cuda", torch.cuda.current_device())
        self.depth = torch.zeros((self.H, self.W), dtype=torch.float32, device=dev)
        self.model_v, self.model_n, self.in_v, self.in_n = (torch.empty((self.H, self.W, 4), dtype=torch.float32, device=dev)
                                                            for _ in range(4))
        self.stream = stream
        self.pose = None
        self.frames = 0

    def start(self, depth_u16, start_pose):
        self.pose = np.asarray(start_pose, np.float64).reshape(4, 4).copy()
        self.table.integrate_depth(self.pose.astype(np.float32), depth_u16, self.k_inv)
        self.table.raycast_maps(self.pose.astype(np.float32), self.depth, self.model_v, self.model_n)
        self.frames = 1
        return self.pose

    def track(self, depth_u16):
        """pre-process + Align only: the new pose (not yet integrated)."""
        from .hashtable import preprocess
        preprocess(depth_u16, self.k_inv, self.in_v, self.in_n, stream=self.stream)
        delta = self.trk.Align(self.in_v, self.model_v, self.model_n).astype(np.float64)
        self.pose = self.pose @ delta
        return self.pose

    def fuse(self, depth_u16):
        """integrate at the current pose + the model's maps for the next Align."""
        p32 = self.pose.astype(np.float32)
        self.table.integrate_depth(p32, depth_u16, self.k_inv)
        self.table.raycast_maps(p32, self.depth, self.model_v, self.model_n)
        self.frames += 1

    def step(self, depth_u16):
        """track + fuse in one library call (vh_fusion_step): the frame's five stages and the pose update without a return to
        Python in between."""
        sys, it = L.IcpSystem(), C.c_int32()
        trk = self.trk
        pose = np.ascontiguousarray(self.pose, np.float64)
        L.check(trk._lib.vh_fusion_step(self.table._h, trk._h, _ptr(depth_u16), _fp(self.k_inv), _fp(trk.K), trk.dist_thres,
                                        trk.max_iters, trk.flags, 0.1, 5.0, _ptr(self.in_v), _ptr(self.in_n), _ptr(self.depth),
                                        _ptr(self.model_v), _ptr(self.model_n), pose.ctypes.data_as(C.POINTER(C.c_double)),
                                        C.byref(sys), C.byref(it)), "vh_fusion_step")
        self.pose = pose
        trk.last, trk.iterations = system_arrays(sys), int(it.value)
        self.frames += 1
        return self.pose

    def close(self):
        self.trk.close()
