"""Which pixels of a Phong / Blinn-Phong row are NaN on one side only (device vs the literal oracle), and from which sample on.
A test-side probe (it calls the oracle, so it lives under tests/); on the GPU box:  python tests/nan_pixels.py c3p"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import golden_cases as gc
from lumillyrender_amd import abi, device
from oracle import binding as oracle

def main():
    key = sys.argv[1]
    name, edit, w, h, spp, integ, seed, gen, rows = gc.STATED_SIZE_CASES[key]
    desc = gc.load_scene(name, edit, w, h)
    p = desc.render_params(spp=spp, seed=seed, integrator=integ)
    scene = device.Scene(desc)
    img = scene.render(p)
    ref = oracle.render(desc, p, mode=oracle.BVH, pad=0.0, fast=True)
    bad = np.argwhere((np.isnan(img) != np.isnan(ref)).any(axis=2) | (np.isinf(img) != np.isinf(ref)).any(axis=2))
    print("one-sided non-finite pixels:", len(bad))
    def one(fn, y, x, n):
        t = (abi.LrTile * 1)(); t[0].x0, t[0].y0, t[0].w, t[0].h = int(x), int(y), 1, 1
        q = desc.render_params(spp=n, seed=seed, integrator=integ)
        return fn(q, t)[y, x]
    dev_fn = lambda q, t: scene.render(q, t, 1)
    orc_fn = lambda q, t: oracle.render(desc, q, t, 1, mode=oracle.BVH, pad=0.0)
    for y, x in bad[:16]:
        out = {"pixel": [int(x), int(y)], "device": [float(v) for v in img[y, x]], "oracle": [float(v) for v in ref[y, x]]}
        for side, fn in (("device", dev_fn), ("oracle", orc_fn)):
            if np.isfinite(one(fn, y, x, spp)).all():
                out[side + "_first_bad_sample"] = None; continue
            lo, hi = 0, spp                                  # invariant: prefix of lo samples finite, prefix of hi samples not
            while hi - lo > 1:
                mid = (lo + hi) // 2
                if np.isfinite(one(fn, y, x, mid)).all(): lo = mid
                else: hi = mid
            out[side + "_first_bad_sample"] = hi - 1
            out[side + "_value_with_it"] = [float(v) for v in one(fn, y, x, hi)]
        print(json.dumps(out), flush=True)
    scene.close()

if __name__ == "__main__":
    main()
