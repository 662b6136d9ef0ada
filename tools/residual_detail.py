"""The rays of fuzz_traversal seeds on which the tree still differs from brute force, with the numbers the culling slack is
built from: |e1||e2|, the float64 determinant, |o - p0|, the reported distances.  usage: residual_detail.py seed [seed ...]"""
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import fuzz_traversal as fz
from lumillyrender_amd import host, device
for seed in [int(a) for a in sys.argv[1:]]:
    text, s, centre, stretch, cam = fz.scene_text(seed)
    desc = host.Description(text=text)
    o, d = fz.rays_for(desc, 200_000, seed, s, centre, cam)
    types, v = fz.rays_for.types, fz.rays_for.v
    ref = None
    for lbvh in (False, True):
        scene = device.Scene(desc, device_bvh=lbvh)
        if ref is None:
            ref = scene.intersect(o, d, brute=True)
        tp, tt = scene.intersect(o, d)
        print(f"seed {seed} {'device' if lbvh else 'host'} tree: size {s:.3g}, tree info {scene.tree_info()}")
        for i in np.nonzero((tp != ref[0]) | (tt != ref[1]))[0]:
            prim, t = int(ref[0][i]), float(ref[1][i])
            oo, dd = o[i].astype(np.float64), d[i].astype(np.float64)
            line = f"  ray {i} class {fz.rays_for.cat[i]}: brute ({prim}, {t!r}) tree ({int(tp[i])}, {float(tt[i])!r}) type {types[prim] if prim >= 0 else -1}"
            if prim >= 0 and types[prim] == 0:
                p = v[prim].reshape(3, 3); e1, e2 = p[1] - p[0], p[2] - p[0]
                pv = np.cross(dd, e2); det = e1 @ pv; tv = oo - p[0]
                A = np.linalg.norm(e1) * np.linalg.norm(e2)
                n = np.cross(e1, e2); cos = abs(dd @ n) / np.linalg.norm(n)
                x = oo + t * dd; lo, hi = p.min(0), p.max(0)
                outside = float(np.max(np.maximum(lo - x, x - hi)))
                texact = (e2 @ np.cross(tv, e1)) / det
                u64, v64 = (tv @ pv) / det, (dd @ np.cross(tv, e1)) / det
                lateral = not (0.0 <= u64 <= 1.0 and v64 >= 0.0 and u64 + v64 <= 1.0)
                kappa = 8 * 2.0 ** -24 * A / 1e-3
                line += f" |e1||e2| {A:.4g} det64 {det:.4g} cos {cos:.3g} |tv| {np.linalg.norm(tv):.5g} t64 {texact:.6g} t32-t64 {t - texact:.4g} outside box by {outside:.4g}; kappa {kappa:.4g} -> slack at this distance {kappa * (np.linalg.norm(tv) + abs(t)):.4g}; float64 barycentrics ({u64:.4f}, {v64:.4f}) -> {'LATERAL: the exact line misses the triangle' if lateral else 'along the ray: the exact line meets the triangle'}"
            print(line)
        scene.close()
