"""Random mesh scenes, the tree against the definition.  usage: fuzz_traversal.py first_seed n_seeds [rays]

Each seed places the 100k-triangle blob mesh under a random transform -- uniform scale from 1e-3 to 1e3, anisotropic
stretch up to 50:1, any rotation, a translation of up to 1e4 object sizes from the origin -- next to a handful of spheres and
quads of unrelated sizes, builds the scene twice (host SAH tree and device LBVH, both collapsed to quantised 4-wide nodes)
and shoots rays at it: random, aimed at the mesh from near and from far, inside the plane of a triangle (|cos| < 1e-3),
axis-parallel, starting on a surface.  Every ray must get the same primitive and the same distance bits from the tree as
from the device's brute force over all primitives (bvh.rs:131-141's definition: every candidate, then the minimum): the
boxes -- padded, quantised to 8 bits, tested in the ray's own frame -- may only prune what no primitive test accepts, at any
scale and offset.

One class of difference is inherent to culling by distance and is reported separately ("outside own box"): Moller-Trumbore
on a triangle seen edge-on from hundreds of triangle sizes away returns a distance whose rounding error exceeds the
triangle's own extent, so the accepted "hit" lies in front of (or behind) the triangle's padded bounding box.  The reference
collects every leaf the ray's line touches and would report that distance; a traversal that skips boxes starting beyond
the closest hit so far cannot see it.  Such rays are counted, checked to really be of that kind (the brute-force hit lies
outside the primitive's own bounds by more than a few ulp), and do not fail the run; anything else does."""
import sys
import numpy as np
sys.path.insert(0, ".")


def scene_text(seed):
    r = np.random.default_rng(seed)
    u = lambda a, b: float(r.uniform(a, b))
    s = 10.0 ** u(-3, 3)                                            # object size
    stretch = [1.0, 1.0, 1.0]
    if r.random() < 0.4:
        stretch[int(r.integers(0, 3))] = 10.0 ** u(0, 1.7)
    off = 10.0 ** u(-1, 4) * s if r.random() < 0.5 else 0.0
    centre = [u(-1, 1) * off, u(-1, 1) * off, u(-1, 1) * off]
    axis = [u(-1, 1), u(-1, 1), u(-1, 1)]
    objs = [f'[[object]]\nmesh = "blob"\nmaterial = "m"\ntransform = [ {{ type = "scale", vector = [{s * stretch[0]:.8g}, {s * stretch[1]:.8g}, {s * stretch[2]:.8g}] }}, '
            f'{{ type = "axis-angle", axis = [{axis[0]:.6g}, {axis[1]:.6g}, {axis[2]:.6g}], angle = {u(0, 360):.6g} }}, '
            f'{{ type = "translate", vector = [{centre[0]:.8g}, {centre[1]:.8g}, {centre[2]:.8g}] }} ]']
    for i in range(int(r.integers(0, 6))):
        c = [centre[k] + u(-4, 4) * s for k in range(3)]
        if r.random() < 0.5:
            objs.append(f'[[object]]\nmesh = "ball{i % 2}"\nmaterial = "m"\ntransform = [ {{ type = "translate", vector = [{c[0]:.8g}, {c[1]:.8g}, {c[2]:.8g}] }} ]')
        else:
            objs.append(f'[[object]]\nmesh = "panel"\nmaterial = "m"\ntransform = [ {{ type = "scale", vector = [{10 ** u(-2, 1) * s:.8g}, 1, {10 ** u(-2, 1) * s:.8g}] }}, '
                        f'{{ type = "axis-angle", axis = [{u(-1, 1):.6g}, {u(-1, 1):.6g}, {u(-1, 1):.6g}], angle = {u(0, 360):.6g} }}, {{ type = "translate", vector = [{c[0]:.8g}, {c[1]:.8g}, {c[2]:.8g}] }} ]')
    # the camera up to ~1500 object sizes away: the boxes' padding is sized for the scene extent INCLUDING the camera
    cam = [centre[0] + u(-1, 1) * s, centre[1] + u(-1, 1) * s, centre[2] - 5 * s * max(stretch) * 10.0 ** (u(0, 2.5) if r.random() < 0.5 else 0.0)]
    text = f'''mesh = [ {{ name = "blob", type = "obj", path = "models/blob/blob.obj" }}, {{ name = "panel", type = "obj", path = "models/simple/quad.obj" }},
  {{ name = "ball0", type = "sphere", radius = {10 ** u(-2, 0.5) * s:.8g} }}, {{ name = "ball1", type = "sphere", radius = {10 ** u(-1, 1) * s:.8g} }} ]
material = [ {{ name = "m", type = "lambert", albedo = [0.5, 0.5, 0.5] }} ]

[renderer]
integrator = "pt"
samples = 1

[film]
output = "hdr"
resolution = [16, 16]

[sky]
type = "uniform"
color = [1, 1, 1]

[camera]
type = "ideal-pinhole"
fov = 40
transform = [ {{ type = "look-at", origin = [{cam[0]:.8g}, {cam[1]:.8g}, {cam[2]:.8g}], target = [{centre[0]:.8g}, {centre[1]:.8g}, {centre[2]:.8g}], up = [0, 1, 0] }} ]

''' + "\n\n".join(objs) + "\n"
    return text, s, centre, stretch, cam


def rays_for(desc, n, seed, s, centre, cam):
    import ctypes as C
    from lumillyrender_amd import abi
    r = np.random.default_rng(seed + 7)
    d = desc.desc
    raw = np.frombuffer(C.string_at(d.prims, d.n_prims * C.sizeof(abi.LrPrimitive)), dtype=np.uint8).reshape(d.n_prims, C.sizeof(abi.LrPrimitive))
    types = raw[:, 0:4].copy().view(np.int32).reshape(-1)
    v = raw[:, 8:44].copy().view(np.float32).reshape(-1, 9).astype(np.float64)
    tris = np.nonzero(types == 0)[0]
    c = np.asarray(centre)
    k = n // 5
    # 1 random around the object, 2 aimed at triangle points from near, 3 from as far as the camera, 4 in-plane grazing, 5 axis-parallel through vertices
    pick = tris[r.integers(0, len(tris), 4 * k)]
    p0, e1, e2 = v[pick, 0:3], v[pick, 3:6] - v[pick, 0:3], v[pick, 6:9] - v[pick, 0:3]
    a, b = r.random(4 * k), r.random(4 * k)
    fl = a + b > 1
    a[fl], b[fl] = 1 - a[fl], 1 - b[fl]
    q = p0 + a[:, None] * e1 + b[:, None] * e2
    o1 = c + r.standard_normal((k, 3)) * 3 * s; d1 = r.standard_normal((k, 3))
    o2 = q[:k] + r.standard_normal((k, 3)) * 0.3 * s; d2 = q[:k] - o2
    if FAR > 0:                                                     # outside the envelope the padding is designed for
        o3 = q[k:2 * k] + (r.standard_normal((k, 3)) * FAR * s)
    else:                                                           # anywhere between the surface and the (possibly far) camera
        o3 = q[k:2 * k] + (np.asarray(cam) - q[k:2 * k]) * r.random((k, 1)) ** 2
    d3 = q[k:2 * k] - o3
    nrm = np.cross(e1[2 * k:3 * k], e2[2 * k:3 * k]); ln = np.linalg.norm(nrm, axis=1, keepdims=True); nrm = nrm / np.maximum(ln, 1e-300)
    ang = r.random(k) * 2 * np.pi
    inpl = np.cos(ang)[:, None] * e1[2 * k:3 * k] + np.sin(ang)[:, None] * e2[2 * k:3 * k]
    inpl /= np.maximum(np.linalg.norm(inpl, axis=1, keepdims=True), 1e-300)
    d4 = inpl + ((r.random(k) * 2 - 1) * 1e-3)[:, None] * nrm
    o4 = q[2 * k:3 * k] - d4 / np.linalg.norm(d4, axis=1, keepdims=True) * (r.random(k) * 2 * s + 1e-3 * s)[:, None]
    d5 = np.eye(3)[r.integers(0, 3, k)] * r.choice([-1.0, 1.0], (k, 1))
    o5 = p0[3 * k:4 * k] - d5 * (r.random(k) * 3 * s)[:, None]      # through a mesh VERTEX along an axis: hits edges and corners of boxes
    o = np.concatenate([o1, o2, o3, o4, o5]); dd = np.concatenate([d1, d2, d3, d4, d5])
    dd = dd / np.maximum(np.linalg.norm(dd, axis=1, keepdims=True), 1e-300)
    cat = np.repeat(np.arange(5), k)
    perm = r.permutation(len(o))
    rays_for.cat = cat[perm]
    rays_for.types, rays_for.v = types, v
    return np.ascontiguousarray(o[perm], dtype=np.float32), np.ascontiguousarray(dd[perm], dtype=np.float32)


def outside_own_box(i, o, d, prim, t):
    """True when the accepted hit o + t d (float64) lies outside the exact bounds of the primitive it was reported for, by
    more than a few f32 ulp of the quantities involved.  Hits inside their primitive's bounds are what the boxes guarantee
    to keep; the padding is there for the slab arithmetic, not for these."""
    types, v = rays_for.types, rays_for.v
    if prim < 0:
        return False
    if types[prim] == 0:
        p = v[prim].reshape(3, 3); lo, hi = p.min(0), p.max(0)
    else:
        lo, hi = v[prim, 0:3] - v[prim, 3], v[prim, 0:3] + v[prim, 3]
    oo = o[i].astype(np.float64)
    tol = 8 * 2.0 ** -23 * max(abs(float(t)), np.abs(oo).max(), np.abs(lo).max(), np.abs(hi).max())
    x = oo + float(t) * d[i].astype(np.float64)
    return bool(np.any(x < lo - tol) or np.any(x > hi + tol))


def run(seed, n_rays=200_000):
    from lumillyrender_amd import host, device
    text, s, centre, stretch, cam = scene_text(seed)
    desc = host.Description(text=text)
    o, d = rays_for(desc, n_rays, seed, s, centre, cam)
    out = []
    ref = None
    n_excused = [0]
    for lbvh in (False, True):
        scene = device.Scene(desc, device_bvh=lbvh)
        ref = scene.intersect(o, d, brute=True)           # the definition on THIS scene's numbering (exact ties: candidate order of the host tree / primitive index)
        tp, tt = scene.intersect(o, d)
        diff = (tp != ref[0]) | (tt != ref[1])
        idx = np.nonzero(diff)[0]
        excused = [i for i in idx if tp[i] >= 0 and tt[i] >= ref[1][i] and outside_own_box(i, o, d, ref[0][i], ref[1][i])]
        n_excused[0] += len(excused)
        idx = np.setdiff1d(idx, np.asarray(excused, dtype=idx.dtype))
        bad = len(idx)
        if bad and VERBOSE:
            print(f"   {'LBVH' if lbvh else 'SAH'}: by ray class {np.bincount(rays_for.cat[idx], minlength=5)} (random, near, far, grazing, axis)")
            for i in idx[:3]:
                print(f"      ray {i} class {rays_for.cat[i]} o {o[i]} d {d[i]}: tree ({tp[i]}, {tt[i]!r}) brute ({ref[0][i]}, {ref[1][i]!r}) |o-c|/s {np.linalg.norm(o[i] - np.asarray(centre)) / s:.3g}")
        out.append(bad)
        scene.close()
    return out, n_excused[0], float((ref[0] >= 0).mean()), desc.desc.n_prims, s, stretch, float(np.linalg.norm(np.asarray(cam) - np.asarray(centre)) / s)


RESIDUAL_SEEDS = (117, 122, 134, 137, 147, 179, 191, 197, 206, 208, 217, 268, 274, 282, 373, 496, 535, 542, 584, 693, 760, 770)


def residual(seed, n_rays=200_000, brute=True):
    """The rays of one seed on which the tree (host SAH and device-built) differs from the device's brute force (`brute`: see device.Scene.intersect), each with the
    geometry of its BRUTE-FORCE hit in float64: barycentrics (u, v) of triangle.rs:76-88, |cos| between ray and triangle normal,
    and how far outside the triangle's own bounds the reported point o + t d lies (in units of the triangle's extent).
    Returns (rows, unexcused): rows = [(tree, ray, prim, u, v, |cos|, outside)], unexcused = differing rays whose brute-force hit is
    INSIDE its primitive's bounds (the guarantee of DESIGN section 2: must be 0)."""
    from lumillyrender_amd import host, device
    text, s, centre, stretch, cam = scene_text(seed)
    desc = host.Description(text=text)
    o, d = rays_for(desc, n_rays, seed, s, centre, cam)
    types, v = rays_for.types, rays_for.v
    rows, unexcused, ref = [], 0, None
    for lbvh in (False, True):
        scene = device.Scene(desc, device_bvh=lbvh)
        ref = scene.intersect(o, d, brute=brute)          # brute=True: every primitive behind its own exact box (the definition); "all": no box at all (rounds 1-5)
        tp, tt = scene.intersect(o, d)
        for i in np.nonzero((tp != ref[0]) | (tt != ref[1]))[0]:
            prim, t = int(ref[0][i]), float(ref[1][i])
            if not (tp[i] >= 0 and tt[i] >= ref[1][i] and outside_own_box(i, o, d, prim, t)) or types[prim] != 0:
                unexcused += 1
                continue
            p = v[prim].reshape(3, 3); e1, e2 = p[1] - p[0], p[2] - p[0]
            oo, dd = o[i].astype(np.float64), d[i].astype(np.float64)
            pv = np.cross(dd, e2); det = e1 @ pv; tv = oo - p[0]; qv = np.cross(tv, e1)
            uu, vv = (tv @ pv) / det, (dd @ qv) / det
            n = np.cross(e1, e2)
            cos = abs(dd @ n) / (np.linalg.norm(n) * np.linalg.norm(dd))
            x = oo + t * dd; lo, hi = p.min(0), p.max(0)
            outside = float(np.max(np.maximum(lo - x, x - hi)) / max(np.max(hi - lo), 1e-300))
            rows.append(("device" if lbvh else "host", int(i), prim, float(uu), float(vv), float(cos), outside))
        scene.close()
    return rows, unexcused


VERBOSE = False
FAR = float(__import__('os').environ.get('FUZZ_FAR', '0'))   # > 0: 'far' origins scattered this many object sizes around, ignoring the camera (outside the design envelope)

if __name__ == "__main__":
    VERBOSE = True
    first, n = int(sys.argv[1]), int(sys.argv[2])
    n_rays = int(sys.argv[3]) if len(sys.argv) > 3 else 200_000
    fails = 0
    for seed in range(first, first + n):
        try:
            bad, excused, hit, n_prims, s, stretch, camdist = run(seed, n_rays)
        except Exception as e:
            print(f"seed {seed}: ERROR {e}")
            fails += 1
            continue
        flag = "" if sum(bad) == 0 else "   <-- TREE DIFFERS FROM BRUTE FORCE"
        fails += sum(bad) != 0
        print(f"seed {seed}: size {s:.3g} stretch {max(stretch):.3g} camera at {camdist:.3g} sizes, prims {n_prims} hit share {hit:.2f}  differing rays: SAH {bad[0]} LBVH {bad[1]}, outside own box {excused}{flag}")
    print("failures:", fails)
    sys.exit(1 if fails else 0)
