"""Synthetic depth input for the voxel-hashing path.

Nothing is shipped but these generators: analytic scenes are ray-cast into a
depth image and converted to the float4 vertex map the path consumes, the way
the reference's pre-processing does it (CameraTrackingUtils.cu:63-73:
vertex = K_inv * (u, v, 1) * depth, w = 1, invalid depth -> z = 0).

  * sphere scenes (numpy, float64 geometry): the two single-frame scenes the
    survey probed against the reference, used by the parity anchors;
  * room scene (torch, runs on CPU or on the GPU): a 6 x 3 x 5 m box room with
    8 seeded spheres / boxes and a closed camera loop, used by configs C2-C5.
"""
from __future__ import annotations

import math

import numpy as np

# common.h:7-10
FX, FY, CX, CY = 517.3, 516.5, 318.6, 255.3


def intrinsics(width: int = 640, height: int = 480):
    """(fx, fy, cx, cy) of common.h scaled with the resolution, as float32 values."""
    sx, sy = np.float32(width) / np.float32(640.0), np.float32(height) / np.float32(480.0)
    return (np.float32(FX) * sx, np.float32(FY) * sy, np.float32(CX) * sx, np.float32(CY) * sy)


def K_matrix(width: int = 640, height: int = 480, transposed: bool = False) -> np.ndarray:
    fx, fy, cx, cy = intrinsics(width, height)
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], np.float32)
    return K.T.copy() if transposed else K


def verts_from_depth(z: np.ndarray, width: int = 640, height: int = 480) -> np.ndarray:
    """float32 [H, W, 4] vertex map from a camera-z depth image (0 = invalid)."""
    fx, fy, cx, cy = intrinsics(width, height)
    z = z.astype(np.float32)
    u, v = np.meshgrid(np.arange(width, dtype=np.float32), np.arange(height, dtype=np.float32))
    kx = (np.float32(1) / fx) * u + (-cx / fx)       # K_inv * (u, v, 1)
    ky = (np.float32(1) / fy) * v + (-cy / fy)
    out = np.zeros((height, width, 4), np.float32)
    out[..., 0] = kx * z
    out[..., 1] = ky * z
    out[..., 2] = z
    out[..., 3] = 1.0
    return out


def sphere_depth(center, radius: float, inside: bool, width: int = 640, height: int = 480) -> np.ndarray:
    """Camera-z depth of a sphere seen from the origin along +z (identity pose)."""
    fx, fy, cx, cy = [np.float64(a) for a in intrinsics(width, height)]
    u, v = np.meshgrid(np.arange(width, dtype=np.float64), np.arange(height, dtype=np.float64))
    dx, dy = (u - cx) / fx, (v - cy) / fy
    c = np.asarray(center, np.float64)
    a = dx * dx + dy * dy + 1.0
    b = -2.0 * (dx * c[0] + dy * c[1] + c[2])
    cc = float((c * c).sum()) - radius * radius
    disc = b * b - 4.0 * a * cc
    ok = disc >= 0
    sq = np.sqrt(np.where(ok, disc, 0.0))
    t = (-b + sq) / (2 * a) if inside else (-b - sq) / (2 * a)
    return np.where(ok & (t > 0), t, 0.0)


def sphere_outside_scene(width: int = 640, height: int = 480) -> np.ndarray:
    """Sphere r = 0.5 m at z = 1.5 m seen from outside (corner pixels invalid)."""
    return verts_from_depth(sphere_depth((0, 0, 1.5), 0.5, False, width, height), width, height)


def sphere_inside_scene(width: int = 640, height: int = 480) -> np.ndarray:
    """Camera at the centre of a sphere of R = 2 m (every pixel valid)."""
    return verts_from_depth(sphere_depth((0, 0, 0), 2.0, True, width, height), width, height)


def yaw_pose(yaw_deg: float, translation=(0.0, 0.0, 0.0)) -> np.ndarray:
    """Camera->world pose: rotation about the y axis plus a translation."""
    a = math.radians(yaw_deg)
    c, s = math.cos(a), math.sin(a)
    T = np.eye(4, dtype=np.float64)
    T[0, 0], T[0, 2], T[2, 0], T[2, 2] = c, s, -s, c
    T[:3, 3] = translation
    return T.astype(np.float32)


# --------------------------------------------------------------------------
# room scene
# --------------------------------------------------------------------------
ROOM_HALF = (3.0, 1.5, 2.5)     # 6 x 3 x 5 m, centred on the origin; +y is down


def room_primitives(seed: int = 1234, count: int = 8):
    """`count` spheres / boxes with seeded positions and sizes inside the room."""
    rng = np.random.RandomState(seed)
    prims = []
    for i in range(count):
        c = np.array([rng.uniform(-2.4, 2.4), rng.uniform(-0.9, 1.1), rng.uniform(-1.9, 1.9)])
        if i % 2 == 0:
            prims.append(("sphere", c, float(rng.uniform(0.2, 0.45))))
        else:
            prims.append(("box", c, rng.uniform(0.15, 0.4, size=3)))
    return prims


def camera_loop(num_frames: int = 500, radius: float = 1.0, height_y: float = 0.0, phase: float = 0.0,
                laps: int = 1) -> np.ndarray:
    """[n, 4, 4] float32 camera->world poses on a closed circle; yaw follows the tangent."""
    poses = np.zeros((num_frames, 4, 4), np.float64)
    for i in range(num_frames):
        th = phase + 2.0 * math.pi * laps * i / num_frames
        pos = np.array([radius * math.cos(th), height_y, radius * math.sin(th)])
        f = np.array([-math.sin(th), 0.0, math.cos(th)])       # forward = tangent = camera +z
        x = np.array([f[2], 0.0, -f[0]])
        y = np.array([0.0, 1.0, 0.0])
        T = np.eye(4)
        T[:3, 0], T[:3, 1], T[:3, 2], T[:3, 3] = x, y, f, pos
        poses[i] = T
    return poses.astype(np.float32)


def render_room_verts(pose, width: int = 640, height: int = 480, prims=None, device="cpu"):
    """Vertex map [H, W, 4] (torch float32 on `device`) of the room seen from `pose`.

    Rays are cast in float32 on `device`; the camera-space ray is (dx, dy, 1) so
    the ray parameter is the camera depth.
    """
    import torch

    if prims is None:
        prims = room_primitives()
    fx, fy, cx, cy = [float(a) for a in intrinsics(width, height)]
    dev = torch.device(device)
    T = torch.as_tensor(np.asarray(pose, np.float32).reshape(4, 4), device=dev)
    u = torch.arange(width, dtype=torch.float32, device=dev)
    v = torch.arange(height, dtype=torch.float32, device=dev)
    kx = ((1.0 / fx) * u + (-cx / fx))[None, :].expand(height, width)
    ky = ((1.0 / fy) * v + (-cy / fy))[:, None].expand(height, width)
    dcam = torch.stack([kx, ky, torch.ones_like(kx)], dim=-1)               # [H, W, 3]
    d = dcam @ T[:3, :3].T                                                   # world direction
    o = T[:3, 3]
    inf = torch.full((height, width), float("inf"), device=dev)

    # room walls: the camera is inside, take the nearest exit
    half = torch.tensor(ROOM_HALF, device=dev)
    t_exit = torch.where(d > 0, (half - o) / d, torch.where(d < 0, (-half - o) / d, inf[..., None]))
    best = t_exit.min(dim=-1).values
    for kind, c, size in prims:
        c_t = torch.tensor(np.asarray(c, np.float32), device=dev)
        if kind == "sphere":
            oc = o - c_t
            a = (d * d).sum(-1)
            b = 2.0 * (d * oc).sum(-1)
            cc = (oc * oc).sum() - float(size) ** 2
            disc = b * b - 4.0 * a * cc
            sq = torch.sqrt(torch.clamp(disc, min=0.0))
            t = (-b - sq) / (2.0 * a)
            t = torch.where((disc >= 0) & (t > 1e-3), t, inf)
        else:
            h = torch.tensor(np.asarray(size, np.float32), device=dev)
            inv = 1.0 / torch.where(d == 0, torch.full_like(d, 1e-30), d)
            t0 = (c_t - h - o) * inv
            t1 = (c_t + h - o) * inv
            tn = torch.minimum(t0, t1).max(dim=-1).values
            tf = torch.maximum(t0, t1).min(dim=-1).values
            t = torch.where((tn <= tf) & (tn > 1e-3), tn, inf)
        best = torch.minimum(best, t)
    z = torch.where(torch.isfinite(best), best, torch.zeros_like(best))
    out = torch.empty((height, width, 4), dtype=torch.float32, device=dev)
    out[..., 0] = kx * z
    out[..., 1] = ky * z
    out[..., 2] = z
    out[..., 3] = 1.0
    return out
