"""Helpers shared by the GPU parity suites (tests/test_gpu_*.py): scene loading, ray generators, the per-pixel bar."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.conftest import ROOT, scene_path  # noqa: F401
from tests import golden_cases as gc  # noqa: F401


TOL = 1e-4

def linf(a, b):
    assert not np.isnan(a).any() and not np.isnan(b).any()
    return float(np.max(np.abs(a - b)))

def load(name, w, h, text_edit=None):
    from lumillyrender_amd import host
    if text_edit is None:
        d = host.Description(scene_path(name))
    else:
        d = host.Description(text=text_edit(open(scene_path(name)).read()))
    d.set_resolution(w, h)
    return d

def _random_rays(desc, n, seed):
    rng = np.random.default_rng(seed)
    prims = desc.dump()["prims"]
    pts = []
    for p in prims:
        v = np.array(p["v"], dtype=np.float32)
        pts.append(v[:3])
    pts = np.array(pts)
    lo, hi = pts.min(0) - 50, pts.max(0) + 50
    o = (rng.random((n, 3)) * (hi - lo) + lo).astype(np.float32)
    tgt = pts[rng.integers(0, len(pts), n)] + rng.standard_normal((n, 3)).astype(np.float32) * 60
    d = (tgt - o).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    # a share of axis-parallel directions (zero components -> infinities in the slab test)
    ax = rng.integers(0, n, n // 16)
    d[ax] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, len(ax))] * rng.choice([-1.0, 1.0], (len(ax), 1)).astype(np.float32)
    return o, d.astype(np.float32)

CASES = [
    # scene, w, h, spp, integrator (None = scene's own)
    ("two-spheres.toml", 48, 48, 32, None),            # spheres + uniform sky, pt
    ("cbox-spheres.toml", 40, 40, 24, 0),              # triangles + spheres, pt (emission through bounces)
    ("cbox-spheres.toml", 40, 40, 24, 1),              # pt-direct: NEE + shadow rays
    ("brdf-row.toml", 64, 36, 32, 1),                  # GGX + Lambert, quad light with mtl material
]

def _generated_assets():
    import os
    from lumillyrender_amd import host
    return os.path.exists(os.path.join(host.ASSET_ROOT, "models/blob/blob.obj")) and os.path.exists(os.path.join(host.ASSET_ROOT, "models/ibl/sky_3k.hdr"))

GLASS = '  { name = "glass", type = "ideal-refraction", reflectance = [0.95, 0.98, 0.95], absorbtance = 0.002, ior = 1.5 },\n'

def _prim_array(desc):
    """(n, 9) float32 vertex data and (n,) types straight from the LrSceneDesc arrays."""
    import ctypes as C
    from lumillyrender_amd import abi
    d = desc.desc
    n = d.n_prims
    raw = np.frombuffer(C.string_at(d.prims, n * C.sizeof(abi.LrPrimitive)), dtype=np.uint8).reshape(n, C.sizeof(abi.LrPrimitive))
    types = raw[:, 0:4].copy().view(np.int32).reshape(n)
    v = raw[:, 8:44].copy().view(np.float32).reshape(n, 9)
    return types, v

def _mesh_rays(desc, n, seed, grazing_share=0.25):
    """Rays through the mesh-box scene; `grazing_share` of them lie (almost) in the plane of a mesh triangle:
    |cos(direction, face normal)| < 1e-3, passing through a point of that triangle."""
    rng = np.random.default_rng(seed)
    types, v = _prim_array(desc)
    tris = np.nonzero(types == 0)[0]
    n_g = int(n * grazing_share)
    o = (rng.random((n, 3)) * [556, 548, 559]).astype(np.float32)
    tgt = (np.array([255, 95, 278]) + rng.standard_normal((n, 3)) * 90).astype(np.float32)
    d = tgt - o
    # grazing part
    pick = tris[rng.integers(0, len(tris), n_g)]
    p0, p1, p2 = v[pick, 0:3].astype(np.float64), v[pick, 3:6].astype(np.float64), v[pick, 6:9].astype(np.float64)
    e1, e2 = p1 - p0, p2 - p0
    nrm = np.cross(e1, e2)
    ok = np.linalg.norm(nrm, axis=1) > 0
    nrm[ok] /= np.linalg.norm(nrm[ok], axis=1, keepdims=True)
    a, b = rng.random(n_g), rng.random(n_g)
    flip = a + b > 1
    a[flip], b[flip] = 1 - a[flip], 1 - b[flip]
    q = p0 + a[:, None] * e1 + b[:, None] * e2                            # a point of the triangle
    ang = rng.random(n_g) * 2 * np.pi
    inplane = np.cos(ang)[:, None] * e1 + np.sin(ang)[:, None] * e2
    inplane /= np.maximum(np.linalg.norm(inplane, axis=1, keepdims=True), 1e-30)
    tilt = (rng.random(n_g) * 2 - 1) * 1e-3
    dg = inplane + tilt[:, None] * nrm
    dg /= np.linalg.norm(dg, axis=1, keepdims=True)
    og = q - dg * (rng.random(n_g) * 60 + 0.5)[:, None]
    o[:n_g], d[:n_g] = og.astype(np.float32), dg.astype(np.float32)
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    cosg = np.abs(np.sum(d[:n_g].astype(np.float64) * nrm, axis=1))
    assert (cosg[ok] < 1.01e-3).all()
    perm = rng.permutation(n)                                             # mix them through the waves
    return np.ascontiguousarray(o[perm]), np.ascontiguousarray(d[perm]), n_g

def _lamp(i, x, z, sx=30, sz=22):
    return (f'[[object]]\nname = "lamp{i}"\nmesh = "panel"\nmaterial = "dark"\ntransform = [\n'
            f'  {{ type = "axis-angle", axis = [0, 0, 1], angle = 180 }},\n  {{ type = "scale", vector = [{sx}, 1, {sz}] }},\n'
            f'  {{ type = "translate", vector = [{x}, 545, {z}] }},\n]\n')

def many_emitter_scene(t, extra_spheres=0):
    """Cornell box with seven quad lights of different sizes (14 emissive triangles) and one emissive sphere:
    15 emitters -> the binary-search branch of sample_emission, and sphere.rs:79-84 as an area light."""
    lamps = [(0, 120, 120, 30, 22), (1, 278, 120, 45, 15), (2, 430, 120, 20, 20), (3, 120, 300, 25, 40),
             (4, 430, 300, 35, 25), (5, 200, 460, 50, 12), (6, 400, 460, 18, 30)]
    t = t.replace('light = [ { type = "area", object = "lamp", emission = [40.0, 30.901960, 22.431360], intensity = 0.7 } ]',
                  "light = [\n" + "".join(f'  {{ type = "area", object = "lamp{i}", emission = [{14 + 3 * i}, {12 + 2 * i}, {9 + i}], intensity = 0.8 }},\n' for i, *_ in lamps)
                  + '  { type = "area", object = "glow", emission = [9, 10, 14] },\n]')
    t = t.replace('  { name = "ball",   type = "sphere", radius = 100 },', '  { name = "ball",   type = "sphere", radius = 100 },\n  { name = "bulb",   type = "sphere", radius = 35 },\n  { name = "bead",   type = "sphere", radius = 9 },')
    head, tail = t.split('[[object]]\nname = "lamp"')
    tail = tail.split("[[object]]", 1)[1]                                   # drop the original lamp object
    objs = "".join(_lamp(*l) for l in lamps)
    objs += '[[object]]\nname = "glow"\nmesh = "bulb"\nmaterial = "dark"\ntransform = [ { type = "translate", vector = [278, 330, 330] } ]\n\n'
    rng = np.random.default_rng(4)
    for k in range(extra_spheres):
        c = rng.random(3) * [440, 250, 400] + [60, 30, 80]
        objs += f'[[object]]\nmesh = "bead"\nmaterial = "matte"\ntransform = [ {{ type = "translate", vector = [{c[0]:.3f}, {c[1]:.3f}, {c[2]:.3f}] }} ]\n\n'
    return head + objs + "[[object]]" + tail

def _ulp_neighbours(centres, k):
    """every float within k ulp of each centre"""
    c = np.asarray(centres, dtype=np.float32)
    bits = c.view(np.int32).astype(np.int64)
    # monotone integer key of a float: negative floats mirror
    key = np.where(bits < 0, -(bits & 0x7fffffff), bits)
    keys = (key[:, None] + np.arange(-k, k + 1)[None, :]).reshape(-1)
    b = np.where(keys < 0, (-keys) | 0x80000000, keys).astype(np.uint32)
    out = b.view(np.float32)
    return out[np.isfinite(out)]

def _counters(st):
    return (st.samples, st.segments, st.shadow_rays, st.sky_fetches)

FLAT_CASES = [("cbox-spheres.toml", 1), ("cbox-spheres.toml", 0), ("brdf-row.toml", 1), ("brdf-row.toml", 0), ("two-spheres.toml", None)]

FULL_SIZE = [("brdf-row.toml", 960, 540), ("mesh-box.toml", 1920, 1370), ("ibl-lens.toml", 2048, 2048)]

def _directions(rng, n):
    d = rng.standard_normal((n, 3))
    axes = np.array([[0, 1, 0], [0, -1, 0], [1, 0, 0], [-1, 0, 0], [0, 0, 1], [0, 0, -1]], dtype=np.float64)
    d = np.concatenate([d, axes])
    return (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)

def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)

def _within_bar(img, ref):
    """L-infinity < 1e-4 per channel on the linear film; a pixel brighter than 1 (HDR films: IBL texels of ~10^3, light sources
    seen directly) gets 1e-4 of ITS OWN value -- per pixel, not 1e-4 of the film's maximum."""
    return np.abs(img - ref) < TOL * np.maximum(1.0, np.abs(ref))

STATED = [pytest.param(c, id=f"{c[0].replace('.toml', '')}-{c[4]}spp") for c in gc.STATED_SPP_CASES]

def _edge_rays(desc, n, seed):
    """Rays aimed at the EDGES of the primitives' own boxes (corners and points along the box edges, +- a few ulp), from origins
    inside the scene's bounds and from the camera: where aabb.rs:74-92 and the primitive tests disagree."""
    rng = np.random.default_rng(seed)
    d = desc.desc
    lo = np.empty((d.n_prims, 3), dtype=np.float64); hi = np.empty((d.n_prims, 3), dtype=np.float64)
    for i in range(d.n_prims):
        p = d.prims[i]
        v = np.array(p.v[:9], dtype=np.float64).reshape(3, 3)
        if p.type == 0:
            lo[i], hi[i] = v.min(axis=0), v.max(axis=0)
        else:
            lo[i], hi[i] = v[0] - p.v[3], v[0] + p.v[3]
    glo, ghi = lo.min(axis=0), hi.max(axis=0)
    k = rng.integers(0, d.n_prims, n)
    f = rng.random((n, 3))
    snap = rng.integers(0, 3, (n, 3))                       # per axis: 0 = lower face, 1 = upper face, 2 = anywhere between
    tgt = np.where(snap == 0, lo[k], np.where(snap == 1, hi[k], lo[k] + f * (hi[k] - lo[k])))
    tgt = tgt * (1.0 + rng.integers(-3, 4, (n, 3)) * 6e-8)  # +- 3 ulp
    cam = np.array(d.camera.aperture_position[:3], dtype=np.float64)
    o = np.where(rng.random((n, 1)) < 0.3, cam[None, :], glo + rng.random((n, 3)) * (ghi - glo))
    dirs = tgt - o
    dirs /= np.maximum(np.linalg.norm(dirs, axis=1, keepdims=True), 1e-30)
    # every 16th ray: axis-parallel, its origin exactly IN planes of the target's box (target snapped to faces, not perturbed): two
    # direction components are zero and (plane - o) is zero on those axes -- the 0 * inf = NaN and +-inf cases of aabb.rs:74-92
    ax = np.arange(n) % 16 == 0
    b = rng.integers(0, 3, n)
    e = np.eye(3)[b] * rng.choice([-1.0, 1.0], (n, 1))
    snapped = np.where(snap == 0, lo[k], np.where(snap == 1, hi[k], lo[k] + f * (hi[k] - lo[k])))
    o = np.where(ax[:, None], snapped - e * (rng.random((n, 1)) * 300.0 + 1.0), o)
    dirs = np.where(ax[:, None], e, dirs)
    return o.astype(np.float32), dirs.astype(np.float32)

STATED_SIZE = sorted(gc.STATED_SIZE_CASES)

def _render_tiles(dev, key, flags=0):
    name, edit, w, h, spp, integ, seed, gen, rows = gc.STATED_SIZE_CASES[key]
    if gen and not gc.have_generated_assets():
        pytest.skip("generated assets missing")
    desc = gc.load_scene(name, edit, w, h)
    tl = gc.stated_tiles(w, h, rows)
    p = desc.render_params(spp=spp, seed=seed, integrator=integ, flags=flags)
    scene = dev.Scene(desc)
    img = scene.render(p, gc.tile_array(tl), len(tl))
    st = scene.stats()
    scene.close()
    return desc, p, tl, img, st

def p_w(desc):
    return int(desc.desc.camera.resolution[0])

def p_h(desc):
    return int(desc.desc.camera.resolution[1])


def usable_cores():
    """Host threads the oracle should start: the affinity mask capped by the cgroup CPU quota (256 threads against a 16-core quota run at half the rate)."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = max(1, min(cores, -(-int(quota) // int(period))))
    except Exception:
        pass
    return cores
