"""Checks on the oracle that do not need the reference: math accuracy vs numpy, sampler/BSDF
consistency, estimator agreement (pt vs pt-direct), traversal modes, determinism, golden fixtures."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from lumillyrender_amd import abi, host
from oracle import binding as oracle
from tests.conftest import ROOT, scene_path

f3 = oracle.f3


def ulp_err(got, ref):
    ref = np.asarray(ref, dtype=np.float64)
    ulp = np.spacing(np.abs(ref).astype(np.float32)).astype(np.float64)
    return float(np.max(np.abs(got.astype(np.float64) - ref) / ulp))


def test_math_accuracy():
    rng = np.random.default_rng(0)
    x = (rng.random(5000) * 2 * np.pi).astype(np.float32)
    assert ulp_err(oracle.math1("sin", x), np.sin(x.astype(np.float64))) < 2.5
    assert ulp_err(oracle.math1("cos", x), np.cos(x.astype(np.float64))) < 2.5
    u = (rng.random(5000) * 2 - 1).astype(np.float32)
    assert ulp_err(oracle.math1("acos", u), np.arccos(u.astype(np.float64))) < 2.5
    a, b = (rng.random(3000) * 4 - 2).astype(np.float32), (rng.random(3000) * 4 - 2).astype(np.float32)
    assert np.max(np.abs(oracle.math2("atan2", a, b) - np.arctan2(a.astype(np.float64), b.astype(np.float64)))) < 5e-7
    base, ex = rng.random(3000).astype(np.float32) + np.float32(1e-3), (rng.random(3000) * 100).astype(np.float32)
    ref = np.power(base.astype(np.float64), ex.astype(np.float64))
    got = oracle.math2("pow", base, ex)
    m = ref > 1e-35
    assert np.max(np.abs(got[m] - ref[m]) / ref[m]) < 1.2e-7
    e = (rng.random(3000) * 40 - 30).astype(np.float32)
    assert ulp_err(oracle.math1("exp", e), np.exp(e.astype(np.float64))) < 1.0
    p = (rng.random(5000) * 5000).astype(np.float32)
    for k in (150.0, 30.0, 300.0, 1.0):
        assert np.array_equal(oracle.math2("fmod_pos", p, np.full_like(p, k)), np.fmod(p, np.float32(k)))


def test_pow_special_cases():
    L = oracle.lib()
    assert np.isnan(L.lr_oracle_pow(-0.5, 2.5))          # the Phong NEE hazard (phong.rs:41-44 with cos < 0)
    assert L.lr_oracle_pow(-2.0, 3.0) == -8.0
    assert L.lr_oracle_pow(0.0, 2.0) == 0.0 and L.lr_oracle_pow(5.0, 0.0) == 1.0
    assert L.lr_oracle_pow(0.0, -1.0) == float("inf")
    assert abs(L.lr_oracle_pow(2.0, -0.5) - 2 ** -0.5) < 1e-7


def test_rng_uniform_and_keyed():
    v = np.stack([oracle.rng_block(9, p, s, b) for p in range(40) for s in range(10) for b in range(5)])
    assert v.min() >= 0.0 and v.max() < 1.0
    assert abs(v.mean() - 0.5) < 0.01 and abs(v.var() - 1 / 12) < 0.005
    assert not np.array_equal(oracle.rng_block(9, 1, 2, 3), oracle.rng_block(10, 1, 2, 3))
    assert not np.array_equal(oracle.rng_block(9, 1, 2, 3), oracle.rng_block(9, 1, 2, 4))
    assert np.array_equal(oracle.rng_block(9, 1, 2, 3), oracle.rng_block(9, 1, 2, 3))


def _mat(t, color, p0=0.0, p1=0.0):
    m = abi.LrMaterial()
    m.type = t
    m.color[:] = color
    m.param[0], m.param[1] = p0, p1
    return m


def _sample(m, out_, n, xi):
    i, pdf = (C.c_float * 3)(), C.c_float()
    oracle.lib().lr_oracle_material_sample(C.byref(m), f3(out_), f3(n), f3(xi), i, C.byref(pdf))
    return np.array(i[:], dtype=np.float64), float(pdf.value)


def _brdf(m, out_, in_, n, pos=(7.0, 0.0, 11.0)):
    r = (C.c_float * 3)()
    oracle.lib().lr_oracle_material_brdf(C.byref(m), f3(out_), f3(in_), f3(n), f3(pos), r)
    return np.array(r[:], dtype=np.float64)


def test_lambert_sampler_matches_its_pdf():
    """lambert.rs:37-55: directions are unit, in the hemisphere of the oriented normal, and
    brdf*cos/pdf == albedo*checker exactly in expectation (cos-weighted importance sampling)."""
    m = _mat(abi.LR_MAT_LAMBERT, [0.5, 0.6, 0.7])
    n = np.array([0.0, 1.0, 0.0])
    out_ = np.array([0.3, 0.8, -0.2]); out_ /= np.linalg.norm(out_)
    rng = np.random.default_rng(3)
    for _ in range(200):
        xi = rng.random(3)
        d, pdf = _sample(m, out_, n, xi)
        assert abs(np.linalg.norm(d) - 1) < 1e-5 and d @ n > 0
        assert abs(pdf - (d @ n) / np.pi) < 1e-6
        w = _brdf(m, out_, d, n) * (d @ n) / pdf
        assert np.allclose(w, np.array([0.5, 0.6, 0.7]) * 1.0, atol=1e-5)      # checker(7, 11) == 1
    # flipped normal: sampling goes to the side the viewer is on, the pdf uses the RAW normal (negative)
    d, pdf = _sample(m, out_, -n, (0.2, 0.4, 0.0))
    assert d @ n > 0 and pdf < 0


def test_checker_levels():
    """lambert.rs:66-90"""
    def ck(u, v):
        r = (C.c_float * 3)()
        oracle.lib().lr_oracle_checker(u, v, r)
        return r[0]
    assert ck(1.0, 75.0) == 0.5          # 2-wide line every 150
    assert ck(31.5 - 1.0, 75.0) == pytest.approx(0.6)   # 1-wide line every 30 (30.5)
    assert ck(75.0, 225.0) == pytest.approx(0.8)        # exactly one of the 150-of-300 bands
    assert ck(75.0, 75.0) == 1.0 and ck(225.0, 225.0) == 1.0
    assert ck(-1.0, 75.0) == pytest.approx(0.8)   # signed_mod(-1, 150) = 150 - 1 = 149: no line; 300-band: 299 vs 75 -> one of two
    assert ck(-149.5, 75.0) == 0.5


def test_ggx_sample_pdf_and_reciprocity():
    """ggx.rs:87-113: pdf = D(h) (h.n) / (4 o.h); brdf symmetric in (in, out)."""
    m = _mat(abi.LR_MAT_GGX, [1, 1, 1], 0.6, 1e5)
    n = np.array([0.0, 0.0, 1.0])
    out_ = np.array([0.4, 0.1, 0.9]); out_ /= np.linalg.norm(out_)
    rng = np.random.default_rng(5)
    for _ in range(100):
        d, pdf = _sample(m, out_, n, rng.random(3))
        assert abs(np.linalg.norm(d) - 1) < 1e-4
        if d @ n > 0.05:
            h = (d + out_); h /= np.linalg.norm(h)
            a2 = (0.6 ** 2) ** 2
            D = a2 / (np.pi * ((a2 - 1) * (h @ n) ** 2 + 1) ** 2)
            assert abs(pdf - D * (h @ n) / (4 * (out_ @ h))) < 1e-4 * max(1, pdf)
            assert np.allclose(_brdf(m, out_, d, n), _brdf(m, d, out_, n), rtol=1e-4)


def test_russian_roulette():
    """scene.rs:64-76"""
    rr = oracle.lib().lr_oracle_russian_roulette
    assert rr(0.7, 0, 5, 64) == 1.0 and rr(0.7, 5, 5, 64) == 1.0
    assert rr(0.7, 6, 5, 64) == pytest.approx(0.7)
    assert rr(0.0, 0, 5, 64) == 0.0                       # black emitter: p = 0 even inside the forced depth
    assert rr(0.8, 66, 5, 64) == pytest.approx(0.8 * 0.25)


def _render(name, w, h, spp, **kw):
    d = host.Description(scene_path(name))
    d.set_resolution(w, h)
    integ = kw.pop("integrator", None)
    p = d.render_params(spp=spp, seed=kw.pop("seed", 1), integrator=integ)
    return oracle.render(d, p, **kw), d, p


def test_traversal_modes_agree():
    """Brute force (the definition) == reference-literal SAH tree + candidate list == padded tree."""
    for name in ("cbox-spheres.toml", "brdf-row.toml"):
        a, _, _ = _render(name, 24, 24, 8)
        b, _, _ = _render(name, 24, 24, 8, mode=oracle.BVH, pad=0.0)
        c, _, _ = _render(name, 24, 24, 8, mode=oracle.BVH, pad=0.05)
        assert np.array_equal(a, b) and np.array_equal(a, c)


def test_threads_do_not_change_the_image():
    a, _, _ = _render("cbox-spheres.toml", 20, 20, 6, threads=1)
    b, _, _ = _render("cbox-spheres.toml", 20, 20, 6, threads=5)
    assert np.array_equal(a, b)


def test_pt_and_pt_direct_agree_in_expectation():
    """scene.rs:153-193: the two estimators integrate the same quantity once depth-0 emission is shown
    (no-direct-emitter = false).  Compare image means within Monte Carlo error."""
    text = open(scene_path("cbox-spheres.toml")).read().replace("no-direct-emitter = true", "no-direct-emitter = false")
    d = host.Description(text=text)
    d.set_resolution(24, 24)
    a = oracle.render(d, d.render_params(spp=600, seed=1, integrator=abi.LR_INTEGRATOR_PT))
    b = oracle.render(d, d.render_params(spp=200, seed=2, integrator=abi.LR_INTEGRATOR_PT_DIRECT))
    ma, mb = a.mean(axis=(0, 1)), b.mean(axis=(0, 1))
    assert np.all(np.abs(ma - mb) / mb < 0.06), (ma, mb)


def test_uniform_sky_furnace_for_black_scene():
    """No emitters, sky (1,1,1), pt: a ray that misses everything returns exactly the sky (sky.rs:17-21)."""
    img, _, _ = _render("two-spheres.toml", 16, 16, 4)
    assert np.all(img[0] == 1.0)                          # top row looks above the horizon


# (the committed golden fixtures -- film crops and per-function vectors -- are pinned in tests/test_golden_fixtures.py)


def test_fuzz_generator_scenes_load_and_render():
    """tools/fuzz_parity.py's random scenes (used by the GPU suite) must stay loadable: three seeds through the host
    loader and the oracle, finite film, both integrators and several cameras among them."""
    import importlib.util, os
    import numpy as np
    from lumillyrender_amd import host
    from oracle import binding as oracle
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(root, "tools", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    seen = set()
    for seed in (0, 1, 5, 17):
        text, integ, cam = fz.scene_text(seed, 16, 12)
        desc = host.Description(text=text)
        desc.set_resolution(16, 12)
        img = oracle.render(desc, desc.render_params(spp=2, seed=seed), threads=1)
        assert img.shape == (12, 16, 3) and np.isfinite(img).all()
        seen.add((integ, cam))
    assert len(seen) >= 3


# ---- round 2: the oracle's own variants agree with each other -----------------------------------------------

def test_fast_build_and_ordered_traversal_give_the_same_bits():
    """oracle/liboracle_fast.so (-O3 -mavx2, what bench.py times as cpu_baseline) and the -O2 parity build produce the
    same film bit for bit, and the 'optimized' baseline mode (ordered early-out walk of the same tree) equals the
    definition (brute force over all primitives)."""
    from lumillyrender_amd import host
    from oracle import binding as oracle
    for name, w, h, spp in (("cbox-spheres.toml", 40, 32, 6), ("brdf-row.toml", 48, 27, 6), ("two-spheres.toml", 32, 32, 8)):
        d = host.Description(os.path.join(ROOT, "scenes", name)); d.set_resolution(w, h)
        p = d.render_params(spp=spp, seed=14)
        brute = oracle.render(d, p, mode=oracle.BRUTE)
        assert np.array_equal(brute, oracle.render(d, p, mode=oracle.BRUTE, fast=True)), name
        assert np.array_equal(brute, oracle.render(d, p, mode=oracle.BVH, pad=0.05, fast=True)), name
        assert np.array_equal(brute, oracle.render(d, p, mode=oracle.BVH_ORDERED, pad=0.05)), name
        assert np.array_equal(brute, oracle.render(d, p, mode=oracle.BVH_ORDERED, pad=0.05, fast=True)), name


def test_math_batch_equals_the_scalar_hooks():
    from oracle import binding as oracle
    rng = np.random.default_rng(4)
    x = (rng.random(500) * 6.28).astype(np.float32)
    assert np.array_equal(oracle.math_batch("sin", x), oracle.math1("sin", x))
    assert np.array_equal(oracle.math_batch("cos", x), oracle.math1("cos", x))
    u = (rng.random(500) * 2 - 1).astype(np.float32)
    assert np.array_equal(oracle.math_batch("acos", u), oracle.math1("acos", u))
    a, b = rng.standard_normal(500).astype(np.float32), rng.standard_normal(500).astype(np.float32)
    assert np.array_equal(oracle.math_batch("atan2", a, b), oracle.math2("atan2", a, b))


def test_threaded_intersect_batch_is_order_independent():
    """lr_oracle_intersect_batch fans out over threads for large batches: same answers as small single-thread batches."""
    from lumillyrender_amd import host
    from oracle import binding as oracle
    d = host.Description(os.path.join(ROOT, "scenes", "cbox-spheres.toml")); d.set_resolution(16, 16)
    rng = np.random.default_rng(6)
    o = (rng.random((6000, 3)) * 500).astype(np.float32)
    v = rng.standard_normal((6000, 3)).astype(np.float32); v /= np.linalg.norm(v, axis=1, keepdims=True)
    p, t = oracle.intersect(d, o, v)
    for lo in range(0, 6000, 1500):
        q, s = oracle.intersect(d, o[lo:lo + 1500], v[lo:lo + 1500])
        assert np.array_equal(p[lo:lo + 1500], q) and np.array_equal(t[lo:lo + 1500], s)


def test_moller_trumbore_distance_error_bound():
    """The culling slack of the 4-wide tree (lumilly_hip.hip Wide4Builder) rests on a bound of the f32 error of
    triangle.rs:69-100's distance: |t32 - t*| <= K eps |e1||e2| (|o - p0| + |t|) / |det| with K = 7.5 from first-order analysis and
    K = 8 in the product.  Checked here against float64 on 2 * 10^6 accepted hits that are aimed at grazing incidence on purpose
    (|cos| down to 1e-4, |det| down to the absolute 1e-3 of triangle.rs:75, distances up to 2000 triangle sizes): the f32 sequence is
    restated in numpy float32 in the reference's operation order (no fused multiply-add) and pinned on a sample against the oracle's
    own triangle test, bit for bit."""
    import ctypes as C
    from oracle import binding as oracle
    rng = np.random.default_rng(77)
    n = 2_000_000
    f32 = np.float32
    size = 10.0 ** rng.uniform(-0.5, 1.7, n)
    p0 = rng.uniform(-300, 300, (n, 3))
    e1 = rng.standard_normal((n, 3)); e1 *= (size * rng.uniform(0.3, 1.0, n) / np.linalg.norm(e1, axis=1))[:, None]
    e2 = rng.standard_normal((n, 3)); e2 *= (size * rng.uniform(0.3, 1.0, n) / np.linalg.norm(e2, axis=1))[:, None]
    nrm = np.cross(e1, e2); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    a, b = rng.random(n), rng.random(n); fl = a + b > 1; a[fl], b[fl] = 1 - a[fl], 1 - b[fl]
    target = p0 + a[:, None] * e1 + b[:, None] * e2
    inpl = np.cos(u := rng.uniform(0, 2 * np.pi, n))[:, None] * e1 / np.linalg.norm(e1, axis=1, keepdims=True) + np.sin(u)[:, None] * e2 / np.linalg.norm(e2, axis=1, keepdims=True)
    cosang = 10.0 ** rng.uniform(-4, 0, n) * rng.choice([-1.0, 1.0], n)
    d = inpl + cosang[:, None] * nrm; d /= np.linalg.norm(d, axis=1, keepdims=True)
    dist = size * 10.0 ** rng.uniform(-1, 3.3, n)
    o = target - d * dist[:, None]
    # the f32 inputs are what both sides see
    P0, E1, E2, O, D = (x.astype(f32) for x in (p0, (p0 + e1).astype(f32) - p0.astype(f32), (p0 + e2).astype(f32) - p0.astype(f32), o, d))

    def cross(x, y):
        return np.stack([x[:, 1] * y[:, 2] - x[:, 2] * y[:, 1], x[:, 2] * y[:, 0] - x[:, 0] * y[:, 2], x[:, 0] * y[:, 1] - x[:, 1] * y[:, 0]], axis=1)

    def dot(x, y):
        return (x[:, 0] * y[:, 0] + x[:, 1] * y[:, 1]) + x[:, 2] * y[:, 2]

    def mt(P0, E1, E2, O, D, dt):
        P0, E1, E2, O, D = (x.astype(dt) for x in (P0, E1, E2, O, D))
        pv = cross(D, E2); det = dot(E1, pv); inv = dt(1.0) / det
        tv = O - P0; uu = dot(tv, pv) * inv; qv = cross(tv, E1); vv = dot(D, qv) * inv; tt = dot(E2, qv) * inv
        ok = ~(np.abs(det) < dt(1e-3)) & ~(uu < 0) & ~(uu > 1) & ~(vv < 0) & ~(uu + vv > 1) & ~(tt < dt(1e-3))
        return tt, det, ok, tv

    t32, det32, ok32, tv32 = mt(P0, E1, E2, O, D, f32)
    t64, det64, _, _ = mt(P0, E1, E2, O, D, np.float64)
    # pin the numpy restatement on the oracle (triangle.rs:69-100 as lr_oracle.cpp states it): same accept / reject, same distance bits
    idx = np.concatenate([np.nonzero(ok32)[0][:300], np.nonzero(~ok32)[0][:100]])
    out = (C.c_float * 7)()
    for i in idx:
        tri = np.concatenate([P0[i], P0[i] + E1[i], P0[i] + E2[i]]).astype(f32)
        if not (np.array_equal((tri[3:6] - tri[0:3]).astype(f32), E1[i]) and np.array_equal((tri[6:9] - tri[0:3]).astype(f32), E2[i])):
            continue                                                       # (the oracle re-derives the edges from the vertices: only exact round trips compare)
        hit = oracle.lib().lr_oracle_triangle_intersect(oracle.f3(tri), oracle.f3(O[i]), oracle.f3(D[i]), 0, out)
        assert bool(hit) == bool(ok32[i]), i
        if hit:
            assert np.float32(out[0]).view(np.uint32) == t32[i].view(np.uint32), (i, out[0], t32[i])
    assert ok32.sum() > 200_000 and (np.abs(det32[ok32]) < 1e-2).sum() > 1_000       # enough accepted hits, enough of them near the threshold
    A = np.linalg.norm(E1.astype(np.float64), axis=1) * np.linalg.norm(E2.astype(np.float64), axis=1)
    scale = 2.0 ** -24 * A * (np.linalg.norm(tv32.astype(np.float64), axis=1) + np.abs(t64)) / np.abs(det64)
    ratio = np.abs(t32.astype(np.float64) - t64)[ok32] / np.maximum(scale[ok32], 1e-300)
    assert float(ratio.max()) <= 7.5, float(ratio.max())                  # the first-order constant holds on every sample (observed maximum: 1.9); the product keeps 8


# ---- round 6: the leaf's own box decides (bvh.rs:20-25), and the candidate order decides exact ties (bvh.rs:38-45,131-141) ----

def _own_box_edge_rays(desc, n, seed):
    """Rays aimed at faces, edges and corners of the primitives' own boxes (+- a few ulp), a third of them from the camera."""
    rng = np.random.default_rng(seed)
    d = desc.desc
    pr = np.ctypeslib.as_array(C.cast(d.prims, C.POINTER(C.c_float)), (d.n_prims, C.sizeof(abi.LrPrimitive) // 4))
    ty = pr[:, 0].view(np.int32)
    # LrPrimitive = {type, material, v[9]} (include/lumilly_hip.h)
    v = pr[:, 2:11].astype(np.float64).reshape(-1, 3, 3)
    tri = (ty == 0)[:, None]
    lo = np.where(tri, v.min(axis=1), v[:, 0] - v[:, 1, :1]); hi = np.where(tri, v.max(axis=1), v[:, 0] + v[:, 1, :1])
    glo, ghi = lo.min(axis=0), hi.max(axis=0)
    k = rng.integers(0, d.n_prims, n)
    f = rng.random((n, 3))
    snap = rng.integers(0, 3, (n, 3))
    tgt = np.where(snap == 0, lo[k], np.where(snap == 1, hi[k], lo[k] + f * (hi[k] - lo[k])))
    tgt = tgt * (1.0 + rng.integers(-3, 4, (n, 3)) * 6e-8)
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


@pytest.mark.parametrize("name,n", [("cbox-spheres.toml", 400_000), ("brdf-row.toml", 400_000), ("two-spheres.toml", 100_000), ("mesh-box.toml", 12_000)])
def test_own_box_definition_equals_the_literal_tree_walk(name, n):
    """The reference's closest hit is tree-independent: min t over the primitives whose OWN exact box passes aabb.rs:74-92 and whose
    own test accepts (mode OWNBOX, a loop over every primitive) == the literal SAH tree + collect-all-candidates walk (mode BVH,
    pad 0) on rays aimed at the faces, edges and corners of the own boxes -- primitive and distance bits, exact ties included (both
    take the first primitive of the candidate order): an inner node's box never rejects what the leaf's box accepts.  And the
    box-free closest hit of rounds 1-5 (mode BRUTE) is NOT the same on such rays."""
    from tests import golden_cases as gc
    if name == "mesh-box.toml" and (not gc.have_generated_assets() or os.environ.get("LUMILLY_TEST_LIGHT")):
        pytest.skip("generated assets missing / the sanitizer run")
    d = host.Description(scene_path(name)); d.set_resolution(16, 16)
    if os.environ.get("LUMILLY_TEST_LIGHT"):
        n = max(2000, n // 20)                  # (the sanitizer run: tests/test_sanitizers.py)
    o, dr = _own_box_edge_rays(d, n, 31)
    p3, t3 = oracle.intersect(d, o, dr, mode=oracle.OWNBOX)
    p1, t1 = oracle.intersect(d, o, dr, mode=oracle.BVH, pad=0.0)
    assert np.array_equal(p1, p3) and np.array_equal(t1.view(np.uint32), t3.view(np.uint32))
    if name != "two-spheres.toml":      # (its radius-1e5 sphere reports distances hundreds of units off its own box: the early out of the ordered walk needs a pad of that size there)
        p5, t5 = oracle.intersect(d, o, dr, mode=oracle.OWNBOX_ORDERED, pad=0.05)
        assert np.array_equal(p5, p3) and np.array_equal(t5.view(np.uint32), t3.view(np.uint32))
    # ties by index: same distances, another primitive at exact ties only
    p7, t7 = oracle.intersect(d, o, dr, mode=oracle.OWNBOX_INDEX)
    p8, t8 = oracle.intersect(d, o, dr, mode=oracle.OWNBOX_TREE_INDEX)
    assert np.array_equal(p7, p8) and np.array_equal(t7.view(np.uint32), t8.view(np.uint32))
    assert np.array_equal(t7.view(np.uint32), t3.view(np.uint32))
    if name in ("cbox-spheres.toml", "brdf-row.toml"):
        p0, t0 = oracle.intersect(d, o, dr, mode=oracle.BRUTE)
        differ = (t0.view(np.uint32) != t3.view(np.uint32)) | ((p0 < 0) != (p3 < 0))
        assert differ.sum() > (50 if n >= 400_000 else 0), int(differ.sum())
        if name == "cbox-spheres.toml" and n >= 400_000:
            assert (p7 != p3).sum() > 0        # the edge rays do produce exact ties between different primitives


def test_literal_render_is_audited_against_the_definition():
    """mode BVH_AUDIT renders through the literal walk and answers every query by the definition as well: no query differs, by a tie
    or otherwise (flat scenes: the definition's loop over every primitive; the mesh: its tie rule on the walk's candidate list)."""
    from tests import golden_cases as gc
    cases = [("cbox-spheres.toml", 40, 40, 24, 1), ("brdf-row.toml", 48, 27, 16, 1)]
    if gc.have_generated_assets() and not os.environ.get("LUMILLY_TEST_LIGHT"):
        cases.append(("mesh-box.toml", 24, 18, 4, 0))
    for name, w, h, spp, integ in cases:
        d = host.Description(scene_path(name)); d.set_resolution(w, h)
        p = d.render_params(spp=spp, seed=2, integrator=integ)
        a, st = oracle.render(d, p, mode=oracle.BVH_AUDIT, with_stats=True)
        assert (st.tie_flips, st.order_dependent) == (0, 0), name
        assert np.array_equal(a.view(np.uint32), oracle.render(d, p, mode=oracle.BVH, pad=0.0).view(np.uint32)), name
        if name != "mesh-box.toml":
            assert np.array_equal(a.view(np.uint32), oracle.render(d, p, mode=oracle.OWNBOX).view(np.uint32)), name
        assert np.array_equal(a.view(np.uint32), oracle.render(d, p, mode=oracle.OWNBOX_ORDERED, pad=0.05, fast=True).view(np.uint32)), name


def test_host_leaf_order_is_the_reference_candidate_order():
    """lr_host_build_bvh's prim_order == the depth-first leaf order of the reference's SAH tree (bvh.rs:69-127 with stable sorts, as the
    oracle builds it), for every max_leaf -- also inside leaves of several primitives.  lr_scene_create numbers the primitives by it,
    which makes the kernels' "lowest id" tie rule bvh.rs:131-141's "first minimum of the candidate list"."""
    from tests import golden_cases as gc
    big = gc.have_generated_assets() and not os.environ.get("LUMILLY_TEST_LIGHT")          # (the sanitizer run keeps the small scenes; the loader suite builds the big trees under ASan)
    names = ["cbox-spheres.toml", "brdf-row.toml", "two-spheres.toml"] + (["mesh-box.toml", "ibl-lens.toml"] if big else [])
    for name in names:
        d = host.Description(scene_path(name))
        n = d.desc.n_prims
        ref = oracle.bvh_leaf_order(d)
        assert sorted(ref.tolist()) == list(range(n))
        assert np.array_equal(np.ctypeslib.as_array(d.desc.bvh_prim_order, (n,)), ref), name
        for ml in (1, 2, 4, 7):
            _, _, order, _ = host.build_bvh(d.desc.prims, n, ml)
            assert np.array_equal(np.array(order[:n], dtype=np.int32), ref), (name, ml)
