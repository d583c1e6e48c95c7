"""Committed golden vectors (tests/golden, produced by make_golden.py from the oracle;
self-pinned, see that script's header).  CPU: the oracle still reproduces them.
GPU: the HIP path reproduces them through the C-ABI."""
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as G  # noqa: E402

KATS = json.load(open(os.path.join(HERE, "golden", "kat_scalars.json")))
SCENES = np.load(os.path.join(HERE, "golden", "scenes.npz"))


def test_scalar_kats(oracle):
    for r in KATS["hash"]:
        assert oracle.hash_block(*r["key"], r["buckets"]) == r["hash"]
    for r in KATS["world2voxel"]:
        p = [float(c) for c in r["p"]]
        assert list(oracle.world2voxel(p, 0.02)) == r["voxel"]
        assert list(oracle.world2block(p, 0.02)) == r["block"]
    for r in KATS["voxel2block"]:
        assert list(oracle.voxel2block(r["v"])) == r["block"]
    for r in KATS["float2int_rz"]:
        assert oracle.float2int_rz(float(r["x"])) == r["i"]
    for r in KATS["invert4x4"]:
        m = np.array([float(c) for c in r["m"]], np.float32)
        assert oracle.invert4x4(m).reshape(-1).view(np.uint32).tolist() == r["inv_bits"]
    from voxelhashing_demo_amd import synth
    KT, K = synth.K_matrix(transposed=True), synth.K_matrix()
    for r in KATS["project"]:
        p = [float(c) for c in r["p"]]
        assert list(oracle.project(KT, p)) == r["kt"] and list(oracle.project(K, p)) == r["k"]


def test_inverse_is_an_inverse(oracle):
    for r in KATS["invert4x4"]:
        m = np.array([float(c) for c in r["m"]], np.float32).reshape(4, 4)
        assert np.allclose(oracle.invert4x4(m) @ m, np.eye(4), atol=1e-5)


def _check(name, table):
    d = G.digest_scene(table)
    for k, v in d.items():
        want = SCENES[f"{name}/{k}"]
        assert np.array_equal(np.asarray(v).reshape(-1), want.reshape(-1)), f"{name}/{k}"


@pytest.mark.parametrize("name", sorted(G.SCENES))
def test_oracle_reproduces_scene(oracle, name):
    class OT(oracle.OracleTable):
        def integrate_np(self, pose, verts, normals=None):
            self.integrate(pose, verts, normals)

        def apply_options(self, opts):
            G.oracle_options(oracle, self, opts)

    t = G.run_scene(name, lambda kw, sem: OT(oracle.default_params(**kw), 640, 480, sem))
    _check(name, t)
    if name == "inside_pin_f2":
        import hashlib
        t.set_raycast_mode(oracle.RAYCAST_FIXED_STEP)
        d = t.raycast(np.eye(4, dtype=np.float32), 0.1, 5.0)
        assert np.array_equal(np.frombuffer(hashlib.sha256(d.tobytes()).digest(), np.uint8),
                              SCENES["raycast_inside_pin_f2/sha"])
        t.set_raycast_mode(oracle.RAYCAST_DDA)
        for jumps in (True, False):
            d, n = t.raycast(np.eye(4, dtype=np.float32), 0.1, 5.0, jumps=jumps, normals=True)
            assert np.array_equal(np.frombuffer(hashlib.sha256(d.tobytes() + n.tobytes()).digest(), np.uint8),
                                  SCENES["raycast_dda_inside_pin_f2/sha"])
        front, back = t.render_blocks(np.eye(4, dtype=np.float32), 0.1, 5.0)
        assert np.array_equal(np.frombuffer(hashlib.sha256(front.tobytes() + back.tobytes()).digest(), np.uint8),
                              SCENES["silhouettes_inside_pin_f2/sha"])
    t.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(G.SCENES))
def test_hip_reproduces_scene(vh, torch_cuda, name):
    torch = torch_cuda

    class GT(vh.SDFHashtable):
        def integrate_np(self, pose, verts, normals=None):
            self.integrate(pose, torch.from_numpy(verts).cuda(), None if normals is None else torch.from_numpy(normals).cuda())
            self.synchronize()

        def apply_options(self, opts):
            G.hip_options(self, opts)

    t = G.run_scene(name, lambda kw, sem: GT(vh.default_params(**kw), 640, 480, sem))
    _check(name, t)
    if name == "inside_pin_f2":
        import hashlib
        d = torch.empty((480, 640), dtype=torch.float32, device="cuda")
        t.set_raycast_mode(vh.RAYCAST_FIXED_STEP)
        t.raycast(np.eye(4, dtype=np.float32), d, 0.1, 5.0)
        t.synchronize()
        assert np.array_equal(d.cpu().numpy()[240].view(np.uint32), SCENES["raycast_inside_pin_f2/row240"])
        t.set_raycast_mode(vh.RAYCAST_DDA)
        n = torch.empty((480, 640, 4), dtype=torch.float32, device="cuda")
        t.raycast_normals(np.eye(4, dtype=np.float32), d, n, 0.1, 5.0)
        t.synchronize()
        dn, nn = d.cpu().numpy(), n.cpu().numpy()
        assert np.array_equal(dn[240].view(np.uint32), SCENES["raycast_dda_inside_pin_f2/row240"])
        assert np.array_equal(nn[240].view(np.uint32), SCENES["raycast_dda_inside_pin_f2/normals_row240"])
        assert np.array_equal(np.frombuffer(hashlib.sha256(dn.tobytes() + nn.tobytes()).digest(), np.uint8),
                              SCENES["raycast_dda_inside_pin_f2/sha"])
        front, back = torch.empty((480, 640), device="cuda"), torch.empty((480, 640), device="cuda")
        t.render_blocks(np.eye(4, dtype=np.float32), front, back, 0.1, 5.0)
        t.synchronize()
        f, b = front.cpu().numpy(), back.cpu().numpy()
        assert np.array_equal(f[240].view(np.uint32), SCENES["silhouettes_inside_pin_f2/front_row240"])
        assert np.array_equal(b[240].view(np.uint32), SCENES["silhouettes_inside_pin_f2/back_row240"])
        assert np.array_equal(np.frombuffer(hashlib.sha256(f.tobytes() + b.tobytes()).digest(), np.uint8),
                              SCENES["silhouettes_inside_pin_f2/sha"])
    t.close()
