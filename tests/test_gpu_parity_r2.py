"""GPU parity, round 2: the holes the first suite left (VERDICT r1 "missing" 4-6, "weak" 1-4).

  * the closest-hit DEFINITION on the device (every ray against every primitive, bvh.rs:131-141 with the candidate set
    widened to the whole scene) against the tree on 10^7 rays of the 100k-triangle scene, a fifth of them grazing a
    mesh face, and against the oracle's brute force on a subset;
  * device branches no test had executed: more than 8 emitters (binary search of objects.rs:37-51), a sphere as the
    area light (sphere.rs:79-84), depth > depth_limit (scene.rs:67-69);
  * sweeps of the deterministic math spec where its argument reduction switches branches;
  * the stand-alone C++ driver (counterpart of main.rs:43-145) and a 2-rank bench.py run on one GPU.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.conftest import ROOT, scene_path

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    from lumillyrender_amd import device
    assert device.device_count() >= 1, "no HIP device: the product path has no CPU fallback"
    return device


@pytest.fixture(scope="module")
def oracle():
    from oracle import binding
    return binding


def load(name, w, h, text_edit=None):
    from lumillyrender_amd import host
    if text_edit is None:
        d = host.Description(scene_path(name))
    else:
        d = host.Description(text=text_edit(open(scene_path(name)).read()))
    d.set_resolution(w, h)
    return d


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


def _generated_assets():
    from lumillyrender_amd import host
    return os.path.exists(os.path.join(host.ASSET_ROOT, "models/blob/blob.obj")) and os.path.exists(os.path.join(host.ASSET_ROOT, "models/ibl/sky_3k.hdr"))


# ---- the definition: brute force over all primitives --------------------------------------------------------

def test_tree_equals_brute_force_on_ten_million_rays(dev, oracle):
    """bvh.rs:131-141 = min over ALL primitives that accept the ray.  The device evaluates exactly that (no boxes at all)
    for 10^7 rays x 10^5 primitives and the 4-wide tree must return the same primitive and the same distance bits --
    including for 2.5 M rays that graze a mesh face at |cos| < 1e-3, where Moeller-Trumbore's rounding is at its
    worst and a box that prunes too eagerly would show (the admitted pruning window)."""
    if not _generated_assets():
        pytest.skip("generated assets missing (run __graft_entry__.build())")
    desc = load("mesh-box.toml", 32, 32)
    scene = dev.Scene(desc)
    n = 10_000_000
    o, d, n_g = _mesh_rays(desc, n, 17)
    assert n_g >= 0.2 * n
    tp, tt = scene.intersect(o, d)
    bp, bt = scene.intersect(o, d, brute=True)
    bad = np.nonzero((tp != bp) | (tt != bt))[0]
    assert bad.size == 0, f"{bad.size} of {n} rays differ, first: ray {bad[:5]}, tree {tp[bad[:5]]} {tt[bad[:5]]}, brute {bp[bad[:5]]} {bt[bad[:5]]}"
    assert (bp >= 0).mean() > 0.9 and (bp < 100000).mean() > 0.2       # most rays hit; a good share hit the mesh itself
    # ... and the device's definition is the oracle's definition (BRUTE mode, plain loop over the primitives)
    m = 40_000
    op, ot = oracle.intersect(desc, o[:m], d[:m], mode=oracle.OWNBOX)
    assert np.array_equal(bp[:m], op) and np.array_equal(bt[:m], ot)
    # the device-built LBVH prunes with different boxes: same answers
    lb = dev.Scene(desc, device_bvh=True)
    lp, lt = lb.intersect(o[:2_000_000], d[:2_000_000])
    assert np.array_equal(lt, bt[:2_000_000]) and (lp != bp[:2_000_000]).mean() < 1e-4      # (another primitive at exact ties only: index order vs the reference's candidate order)
    scene.close(); lb.close()


@pytest.mark.parametrize("seed", [0, 3, 4, 17, 79, 121])
def test_tree_equals_brute_force_under_random_transforms(dev, seed):
    """tools/fuzz_traversal.py's scenes: the mesh scaled 1e-3..1e3, stretched up to 50:1, rotated, moved up to 1e4 sizes
    off the origin, the camera up to ~1e4 sizes away, rays from anywhere between surface and camera, grazing and
    axis-parallel through vertices.  Host SAH tree and device LBVH must both return brute force's primitive and distance
    bits for EVERY ray (a Moeller-Trumbore distance that lands outside the triangle's own box -- edge-on
    slivers seen from far away -- is what distance culling cannot reproduce; the tree therefore does not cull by distance above
    such triangles.  Seeds 4, 17, 79 and 121 contain such rays)."""
    if not _generated_assets():
        pytest.skip("generated assets missing (run __graft_entry__.build())")
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_traversal", os.path.join(ROOT, "tools", "fuzz_traversal.py"))
    ft = importlib.util.module_from_spec(spec); spec.loader.exec_module(ft)
    bad, excused, hit, n_prims, *_ = ft.run(seed, 100_000)
    assert bad == [0, 0], f"seed {seed}: {bad} rays differ from brute force inside their primitive's own bounds"
    # round 3: nodes above a sliver triangle (sin of the angle at p0 < 1/8) are exempt from distance culling, so the rays
    # round 2 had to excuse (seeds 4, 17, 79, 121) now get brute force's answer too: bvh.rs:131-141 holds for every ray
    assert excused == 0, f"seed {seed}: {excused} rays whose brute-force hit lies outside its own primitive's box still differ"


@pytest.mark.parametrize("name", ["cbox-spheres.toml", "brdf-row.toml"])
def test_brute_force_kernel_on_flat_scenes(dev, oracle, name):
    """The same three-way agreement on the flat-loop scenes (spheres included)."""
    from tests.test_gpu_parity import _random_rays
    desc = load(name, 32, 32)
    scene = dev.Scene(desc)
    o, d = _random_rays(desc, 200_000, 23)
    tp, tt = scene.intersect(o, d)
    bp, bt = scene.intersect(o, d, brute=True)
    op, ot = oracle.intersect(desc, o, d, mode=oracle.OWNBOX)
    assert np.array_equal(tp, bp) and np.array_equal(tt, bt)
    assert np.array_equal(bp, op) and np.array_equal(bt, ot)
    scene.close()


def test_mesh_film_against_brute_force_oracle(dev, oracle):
    """the mesh image parity ran tree-vs-tree.  Here the oracle renders the 100k-triangle scene in
    BRUTE mode (every ray against every primitive) on a film small enough for that."""
    if not _generated_assets():
        pytest.skip("generated assets missing")
    desc = load("mesh-box.toml", 20, 15)
    params = desc.render_params(spp=2, seed=33)
    scene = dev.Scene(desc)
    img = scene.render(params)
    ref, ost = oracle.render(desc, params, mode=oracle.OWNBOX, with_stats=True)
    st = scene.stats()
    assert (st.samples, st.segments, st.shadow_rays) == (ost.samples, ost.segments, ost.shadow_rays)
    assert float(np.max(np.abs(img - ref))) < TOL
    scene.close()


def test_device_built_bvh_matches_the_oracle(dev, oracle, monkeypatch, capfd):
    """the LBVH test compared HIP with HIP.  Device-built tree vs the oracle, flat and mesh scene.
    LR_DEBUG makes lr_scene_create verify its own radix sort of the Morton codes (hand-written, four 8-bit passes)."""
    from lumillyrender_amd import abi
    monkeypatch.setenv("LR_DEBUG", "1")
    for name, w, h, spp, mode, pad, flags in [("cbox-spheres.toml", 48, 40, 8, oracle.OWNBOX, 0.0, abi.LR_FLAG_STREAMING),
                                              ("mesh-box.toml", 40, 30, 4, oracle.OWNBOX_TREE, 0.0, 0)]:
        if name == "mesh-box.toml" and not _generated_assets():
            continue
        desc = load(name, w, h)
        sc = dev.Scene(desc, device_bvh=True)
        assert sc.stats().bvh_build_ms > 0.0
        assert "radix sort of %d codes verified" % desc.desc.n_prims in capfd.readouterr().err
        p = desc.render_params(spp=spp, seed=19, flags=flags)
        img = sc.render(p)
        ref, ost = oracle.render(desc, desc.render_params(spp=spp, seed=19), mode=mode, pad=pad, with_stats=True)
        st = sc.stats()
        assert (st.segments, st.shadow_rays) == (ost.segments, ost.shadow_rays)
        assert float(np.max(np.abs(img - ref))) < TOL
        sc.close()


# ---- branches that had never run ------------------------------------------------------------------------------

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


@pytest.mark.parametrize("extra", [0, 24])
def test_many_emitters_and_a_sphere_light(dev, oracle, extra):
    """objects.rs:37-51 with 15 emitters (device: binary search over the running area sums), one of them a sphere
    (sphere.rs:79-84 sampled as a light, its emission seen through shadow rays that end on a sphere).  extra = 0 keeps
    the scene flat (27 primitives), extra = 24 pushes it onto the tree; every pipeline (the fused one stages the emitter rows in LDS)."""
    from lumillyrender_amd import abi
    desc = load("cbox-spheres.toml", 44, 40, text_edit=lambda t: many_emitter_scene(t, extra))
    assert desc.desc.n_prims == 10 + 14 + 1 + 2 + extra
    scene = dev.Scene(desc)
    # the pick itself, bit for bit, including draws at and next to both ends
    rng = np.random.default_rng(2)
    xi = np.concatenate([rng.random(200_000).astype(np.float32), np.array([0.0, np.nextafter(np.float32(1), np.float32(0)), 0.5], dtype=np.float32),
                         np.linspace(0, 1, 4097, dtype=np.float32)[:-1]])
    want, n_em = oracle.emitter_pick(desc, xi)
    assert n_em == 15
    got = scene.emitter_pick(xi)
    assert np.array_equal(got, want)
    assert set(np.unique(got)) == set(range(15))
    for integ in (abi.LR_INTEGRATOR_PT_DIRECT, abi.LR_INTEGRATOR_PT):
        ref, ost = oracle.render(desc, desc.render_params(spp=16, seed=12, integrator=integ), with_stats=True)
        for flags in (0, abi.LR_FLAG_STREAMING, abi.LR_FLAG_RESIDENT, abi.LR_FLAG_FUSED):
            img = scene.render(desc.render_params(spp=16, seed=12, integrator=integ, flags=flags))
            st = scene.stats()
            assert (st.samples, st.segments, st.shadow_rays) == (ost.samples, ost.segments, ost.shadow_rays), (integ, flags)
            assert float(np.max(np.abs(img - ref))) < TOL, (integ, flags)
    assert ref.max() > 0.05
    scene.close()


@pytest.mark.parametrize("depth,limit", [(1, 2), (0, 0), (2, 3)])
def test_depth_limit_halving(dev, oracle, depth, limit):
    """scene.rs:64-76: beyond depth_limit the survival probability is halved per bounce (p *= 0.5^(d - limit)); with the
    default limit of 64 no test path ever got there.  depth = 1, depth-limit = 2 puts most vertices on that branch."""
    from lumillyrender_amd import abi

    def edit(t):
        return t.replace("depth = 5", f"depth = {depth}").replace("depth-limit = 64", f"depth-limit = {limit}")
    for name, integ in (("cbox-spheres.toml", 1), ("cbox-spheres.toml", 0), ("brdf-row.toml", 1)):
        text = open(scene_path(name)).read()
        if "depth = 5" not in text:
            edit_fn = lambda t: t.replace("[renderer]\n", f"[renderer]\ndepth = {depth}\ndepth-limit = {limit}\n")
        else:
            edit_fn = edit
        desc = load(name, 40, 30, text_edit=edit_fn)
        r = desc.renderer
        assert (r.depth, r.depth_limit) == (depth, limit)
        scene = dev.Scene(desc)
        ref, ost = oracle.render(desc, desc.render_params(spp=32, seed=3, integrator=integ), with_stats=True)
        for flags in (0, abi.LR_FLAG_STREAMING, abi.LR_FLAG_FUSED):
            img = scene.render(desc.render_params(spp=32, seed=3, integrator=integ, flags=flags))
            st = scene.stats()
            assert (st.segments, st.shadow_rays) == (ost.segments, ost.shadow_rays), (name, integ, flags)
            assert float(np.max(np.abs(img - ref))) < TOL
        # the branch really ran: paths are much shorter than with the default (5, 64)
        assert ost.segments < 4.2 * ost.samples
        scene.close()


# ---- the deterministic math spec where it switches branches ------------------------------------------------------

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


def test_math_spec_sweeps(dev, oracle):
    """device and oracle share one hand-written spec, so sample it where it is fragile.
    acos at both ends of [-1, 1] (the asin-based form switches at |x| = 0.5 and loses digits near 1); sin / cos at every
    multiple of pi/4 up to 2 pi +- 64 ulp (Cody-Waite quadrant changes) and over the whole range of r1 = 2 pi xi;
    atan2 on and next to the axes.  Device == oracle bit for bit, and both within 2 ulp of numpy's float64 result."""
    def ulp_err(got, ref64):
        ref32 = ref64.astype(np.float32)
        ulp = np.spacing(np.abs(ref32)).astype(np.float64)
        return np.abs(got.astype(np.float64) - ref64) / np.maximum(ulp, 1e-45)
    # acos
    x = np.concatenate([_ulp_neighbours([-1.0, 1.0, -0.5, 0.5, 0.0], 1000), np.linspace(-1, 1, 200001, dtype=np.float32)])
    x = x[np.abs(x) <= 1.0]
    g, w = dev.selftest_math(2, x), oracle.math_batch("acos", x)
    assert np.array_equal(g, w)
    assert ulp_err(g, np.arccos(x.astype(np.float64))).max() <= 2.0
    # sin / cos
    k = np.arange(0, 9, dtype=np.float64) * (np.pi / 4)
    a = np.concatenate([_ulp_neighbours(k.astype(np.float32), 64), (np.float32(2 * np.pi) * np.linspace(0, 1, 400001, dtype=np.float32)[:-1]),
                        _ulp_neighbours([np.float32(2 * np.pi)], 64)])
    a = a[(a >= 0) & (a <= np.float32(6.2832))]
    for fn, name, ref in ((0, "sin", np.sin), (1, "cos", np.cos)):
        g, w = dev.selftest_math(fn, a), oracle.math_batch(name, a)
        assert np.array_equal(g, w), name
        r = ref(a.astype(np.float64))
        big = np.abs(r) > 1e-3                          # near a zero of the function the error is absolute, not relative
        assert ulp_err(g[big], r[big]).max() <= 2.0, name
        assert np.abs(g[~big].astype(np.float64) - r[~big]).max() < 2e-7, name
    # atan2: axes and their neighbourhoods, all sign combinations, tiny and huge magnitudes
    mags = np.array([0.0, 1e-30, 1e-6, 0.5, 1.0, 3.0, 1e6, 1e30], dtype=np.float32)
    vals = np.concatenate([mags, -mags, _ulp_neighbours([1.0, -1.0], 4)])
    yy, xx = np.meshgrid(vals, vals)
    yy, xx = yy.reshape(-1).astype(np.float32), xx.reshape(-1).astype(np.float32)
    rng = np.random.default_rng(9)
    dirs = rng.standard_normal((200000, 2)).astype(np.float32)
    yy, xx = np.concatenate([yy, dirs[:, 0], dirs[:, 0] * 1e-5]), np.concatenate([xx, dirs[:, 1], dirs[:, 1]])
    g, w = dev.selftest_math(3, yy, xx), oracle.math_batch("atan2", yy, xx)
    assert np.array_equal(g.view(np.uint32), w.view(np.uint32))
    ok = (yy != 0) | (xx != 0)
    r = np.arctan2(yy.astype(np.float64), xx.astype(np.float64))
    # compared modulo 2 pi: on the negative x axis the spec returns +pi for y = -0.0 where IEEE atan2 returns -pi; sky.rs:60-61
    # maps both to the same texel column ((phi + pi) / 2 pi mod 1 = 0)
    dphi = np.abs(g[ok].astype(np.float64) - r[ok])
    assert np.minimum(dphi, np.abs(dphi - 2 * np.pi)).max() < 1e-6


def test_ibl_texel_lookup_is_exact(dev, oracle):
    """sky.rs:57-78 on the device against the oracle, bit for bit: which texel a miss reads is a discrete decision
    (acos / atan2 / floor), so the IBL image tolerance -- stated RELATIVE to the film's range in test_mesh_scene_parity,
    because the map holds texels of ~1e3 and f32 sums of them carry an ulp of 6e-5 -- never hides a wrong texel.
    Directions: random, the poles, the +-x / +-z axes (atan2 branch cuts, the u = 0 / 1 seam) and their neighbours."""
    if not _generated_assets():
        pytest.skip("generated assets missing")
    desc = load("ibl-lens.toml", 16, 16)
    scene = dev.Scene(desc)
    rng = np.random.default_rng(31)
    d = rng.standard_normal((400_000, 3))
    axes = np.array([[0, 1, 0], [0, -1, 0], [1, 0, 0], [-1, 0, 0], [0, 0, 1], [0, 0, -1]], dtype=np.float64)
    near = (axes[:, None, :] + rng.standard_normal((6, 4000, 3)) * 1e-4).reshape(-1, 3)
    seam = np.stack([-np.ones(4000), rng.uniform(-1, 1, 4000), rng.standard_normal(4000) * 1e-6], axis=1)   # phi = +-pi: the u seam
    d = np.concatenate([d, axes, near, seam])
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    got, want = scene.sky(d), oracle.sky_batch(desc, d)
    assert np.array_equal(got, want)
    assert len(np.unique(got[:, 0])) > 1000 and got.max() > 100.0     # really the HDR map, bright texels included
    scene.close()


# ---- the C++ driver and the multi-rank bench path ------------------------------------------------------------------

def test_standalone_driver_writes_the_same_png(dev, tmp_path):
    """lumilly_render (host/main.cpp), the counterpart of the reference binary (main.rs:43-145): one host thread per GPU,
    lr_render into a shared film, png through the host writer.  Same bytes as the Python path + lr_film_quantize."""
    from lumillyrender_amd import host
    exe = os.path.join(ROOT, "lumillyrender_amd", "lumilly_render")
    if not os.path.exists(exe):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "lumillyrender_amd", "host"), "driver"], check=True)
    out = tmp_path / "driver.png"
    r = subprocess.run([exe, scene_path("cbox-spheres.toml"), "--gpus", "1", "--spp", "8", "--seed", "5", "--out", str(out), "--assets", os.path.join(ROOT, "assets")],
                       capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr + r.stdout
    assert "Msamples/s" in r.stdout and "elapse" in r.stdout
    desc = host.Description(scene_path("cbox-spheres.toml"))
    scene = dev.Scene(desc)
    film = scene.render(desc.render_params(spp=8, seed=5))
    q = scene.quantize("rgb8", gamma=desc.film.gamma)
    from PIL import Image
    got = np.array(Image.open(out).convert("RGB"))
    assert got.shape == q.shape
    # the driver quantises on the host (powf), lr_film_quantize on the device (det_pow): bucket-edge pixels may differ by one
    diff = np.abs(got.astype(int) - q.astype(int))
    assert diff.max() <= 1 and (diff != 0).mean() < 2e-3
    assert np.array_equal(got, host.to_color(film, desc.film.gamma))        # and exactly the host writer applied to the film
    # no scene file / a missing file: an error message and a non-zero exit code, not a panic
    assert subprocess.run([exe], capture_output=True).returncode == 2
    assert subprocess.run([exe, "/nonexistent.toml"], capture_output=True, cwd=ROOT).returncode == 1
    scene.close()


@pytest.mark.parametrize("config", ["c2", "c4"])
def test_two_rank_bench_assembles_the_single_rank_film(tmp_path, config):
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, one process per rank), both ranks on this
    box's one GPU: the film assembled in host shared memory equals the 1-rank film bit for bit, and the JSON line is the
    stated workload (strong scaling by default).  c2 = the resident pipeline, c4 = the streaming one (mesh scene)."""
    import json
    if config == "c4" and not _generated_assets():
        pytest.skip("generated assets missing (run __graft_entry__.build())")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--config", config, "--steps", "1", "--warmup", "0", "--width", "192", "--height", "128", "--spp", "16", "--no-cpu-baseline", "--tile", "32"]
    one = tmp_path / "one.npy"
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dump-film", str(one)] + common,
                        capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert r1.returncode == 0, r1.stderr[-2000:]
    two = tmp_path / "two.npy"
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                         "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--same-device", "--backend", "gloo",
                         "--dump-film", str(two)] + common, capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert r2.returncode == 0, r2.stderr[-2000:]
    a, b = np.load(one), np.load(two)
    assert np.array_equal(a, b) and a.max() > 0
    line = json.loads([l for l in r2.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["config"]["spp"] == 16
    assert line["rank_render_ms"]["max"] >= line["rank_render_ms"]["min"] > 0


# ---- switches that must not change a film; boundary checks added in round 2 ----------------------------------------

def test_sorted_and_unordered_variants_are_bit_identical(dev, knobs):
    """LR_DENSE=0 (per-class lists + one k_shade launch per class instead of k_shade_all over the slots), LR_SORT=1 (rays
    binned by octant / origin cell before trace and shadow), LR_SHADE_ORDER=0 (lists shaded in list order) and the number
    of slot groups only change which lane handles which ray or vertex: same films, same counters."""
    monkeypatch = knobs                                  # (the knob build of the library: the product one reads no LR_* switch, csrc/lr_knobs.h)
    if not _generated_assets():
        pytest.skip("generated assets missing")
    from lumillyrender_amd import abi
    for name, spp in (("mesh-box.toml", 24), ("ibl-lens.toml", 16), ("brdf-row.toml", 16)):
        desc = load(name, 160, 120)                     # 19200 pixels: several ranges, so the sort window is exercised with real lists
        scene = dev.Scene(desc)
        p = desc.render_params(spp=spp, seed=41, flags=abi.LR_FLAG_STREAMING)
        base = scene.render(p)
        st0 = scene.stats()
        for env in ({"LR_DENSE": "0"}, {"LR_SORT": "1"}, {"LR_DENSE": "0", "LR_SORT": "1"}, {"LR_DENSE": "0", "LR_SHADE_ORDER": "0"},
                    {"LR_DENSE": "0", "LR_SORT": "1", "LR_SHADE_ORDER": "0", "LR_MAXGROUP": "2"}, {"LR_GROUPS": "1"}, {"LR_GROUPS": "3"},
                    {"LR_DENSE": "0", "LR_GROUPS": "1"}):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            img = scene.render(p)
            st = scene.stats()
            for k in env:
                monkeypatch.delenv(k)
            assert np.array_equal(img, base), (name, env)
            assert (st.samples, st.segments, st.shadow_rays, st.sky_fetches) == (st0.samples, st0.segments, st0.shadow_rays, st0.sky_fetches), (name, env)
        scene.close()


def test_resident_workgroup_sizes_are_bit_identical(dev, oracle, knobs):
    """The resident kernel runs 256- or 512-slot workgroups (the host picks 512 for flat scenes with several BSDF lists):
    slots, chunking and RNG keys do not depend on it, so the films are the same bits -- and match the oracle."""
    monkeypatch = knobs                                  # (the knob build of the library: the product one reads no LR_* switch, csrc/lr_knobs.h)
    from lumillyrender_amd import abi
    for name, integ in (("brdf-row.toml", 1), ("cbox-spheres.toml", 1), ("two-spheres.toml", None)):
        desc = load(name, 72, 40)
        scene = dev.Scene(desc)
        films = []
        for rb in ("256", "512"):
            monkeypatch.setenv("LR_RES_BLOCK", rb)
            films.append(scene.render(desc.render_params(spp=24, seed=17, integrator=integ, flags=abi.LR_FLAG_RESIDENT)))
            assert scene.stats().pipeline == 1
        monkeypatch.delenv("LR_RES_BLOCK")
        assert np.array_equal(films[0], films[1]), name
        ref = oracle.render(desc, desc.render_params(spp=24, seed=17, integrator=integ))
        assert float(np.max(np.abs(films[1] - ref))) < TOL
        scene.close()


def test_round2_boundary_checks(dev):
    """a BVH that is a DAG, overlapping tiles, a film of the wrong shape / dtype / layout are refused with an
    error code or a ValueError instead of being expanded exponentially, rendered twice or written out of bounds."""
    import ctypes as C
    from lumillyrender_amd import abi, host
    desc = load("cbox-spheres.toml", 32, 24)
    d = desc.desc
    lib = dev.lib()
    # two parents for one node: the root's second child slot is pointed at a grandchild that node `a` already owns
    nodes = (abi.LrBvhNode * d.n_bvh_nodes)()
    C.memmove(nodes, d.bvh_nodes, C.sizeof(nodes))
    a, ca = next((i, nodes[i].child[c]) for i in range(1, d.n_bvh_nodes) for c in range(2) if nodes[i].child[c] >= 0)
    assert ca > a >= 1
    nodes[0].child[1] = ca
    bad = abi.LrSceneDesc.from_buffer_copy(d)
    bad.bvh_nodes = C.cast(nodes, C.POINTER(abi.LrBvhNode))
    h = C.c_void_p()
    assert lib.lr_scene_create(0, C.byref(bad), C.byref(h)) == abi.LR_EINVAL
    assert b"two parents" in lib.lr_last_error()
    # a node nobody references
    C.memmove(nodes, d.bvh_nodes, C.sizeof(nodes))
    leaf = next(nodes[i].child[c] for i in range(d.n_bvh_nodes) for c in range(2) if nodes[i].child[c] < 0)
    nodes[a].child[0 if nodes[a].child[0] == ca else 1] = leaf
    assert lib.lr_scene_create(0, C.byref(bad), C.byref(h)) == abi.LR_EINVAL
    scene = dev.Scene(desc)
    params = desc.render_params(spp=2, seed=1)
    # overlapping tiles
    tiles = (abi.LrTile * 2)()
    tiles[0].x0, tiles[0].y0, tiles[0].w, tiles[0].h = 0, 0, 20, 12
    tiles[1].x0, tiles[1].y0, tiles[1].w, tiles[1].h = 19, 11, 5, 5
    with pytest.raises(host.LumillyError, match="overlap"):
        scene.render(params, tiles, 2)
    tiles[1].x0 = 20                                                   # abutting is fine
    scene.render(params, tiles, 2)
    # films the native side must never see
    with pytest.raises(ValueError):
        scene.render(params, out=np.zeros((24, 32, 3), dtype=np.float64))
    with pytest.raises(ValueError):
        scene.render(params, out=np.zeros((32, 24, 3), dtype=np.float32))
    with pytest.raises(ValueError):
        scene.render(params, out=np.zeros((24, 64, 3), dtype=np.float32)[:, ::2])
    with pytest.raises(ValueError):
        scene.render(params, tiles, 3)                                  # more tiles than the array holds
    with pytest.raises(ValueError):
        scene.intersect(np.zeros((4, 3), np.float32), np.zeros((5, 3), np.float32))
    assert np.isfinite(scene.render(params)).all()
    scene.close()
