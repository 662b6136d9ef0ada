"""Writes the golden film crops under tests/golden/ with the CPU oracle.

The reference cannot run here (no Rust toolchain, unseeded RNG), so these vectors are the oracle's own
output at a fixed seed: they pin the oracle against drift and give the GPU tests a committed target.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from lumillyrender_amd import host  # noqa: E402
from oracle import binding as oracle  # noqa: E402
from tests.test_oracle_properties import GOLDEN_CASES, golden_name  # noqa: E402

for case in GOLDEN_CASES:
    name, w, h, spp, integ, seed = case
    d = host.Description(os.path.join(ROOT, "scenes", name))
    d.set_resolution(w, h)
    img = oracle.render(d, d.render_params(spp=spp, seed=seed, integrator=integ))
    path = os.path.join(ROOT, "tests", "golden", golden_name(case))
    np.save(path, img)
    print(path, img.shape, float(img.mean()))
