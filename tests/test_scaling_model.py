"""profiles/r05_scaling_model.json is what tools/scaling_model.py makes of the committed inputs (profiles/r05_scaling_inputs_*.json,
measured one rank at a time on one MI355X) -- the prediction bench.py prints in the N-rank line must be reproducible from them --
and dist.scaling_prediction finds the entry of every world size the driver runs."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_committed_model_follows_from_committed_inputs():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "scaling_model.py"),
                          os.path.join(ROOT, "profiles", "r05_scaling_inputs_C2.json"),
                          os.path.join(ROOT, "profiles", "r05_scaling_inputs_C5.json")], capture_output=True, text=True, check=True).stdout
    assert json.loads(out) == json.load(open(os.path.join(ROOT, "profiles", "r05_scaling_model.json")))


def test_every_world_size_has_a_prediction():
    from voxelhashing_demo_amd import dist
    for wl in ("C2", "C5"):
        rates = []
        for n in (1, 2, 4, 8):
            p = dist.scaling_prediction(wl, n, 8)
            assert p and p["reference_walk"]["nominal"]["frames_per_s"] > 0 and p["walk_free"]["nominal"]["frames_per_s"] > 0
            assert p["reference_walk"]["conservative"]["frames_per_s"] <= p["reference_walk"]["nominal"]["frames_per_s"]
            assert p["reference_walk"]["nominal"]["bound"] in ("apply", "collectives", "generation")
            rates.append(p["reference_walk"]["nominal"]["frames_per_s"])
        assert rates == sorted(rates)                       # weak scaling over a fixed table: more ranks, more frames/s
    assert dist.scaling_prediction("C2", 3, 8) is None      # (no entry: the model has 1, 2, 4, 8)
