"""GPU parity -- math.

Device functions of the path against the oracle's deterministic spec, bit for bit: the math spec (sin / cos / acos / atan2 / pow / fmod, the Lambert grid), the
counter-based generator, the exact reciprocal of the triangle test, the IBL texel lookup and its RGBE storage, the five BSDFs, cameras and emitter sampling through
the golden function vectors, GGX at grazing incidence.

(Regrouped by component in round 6; the tests themselves are unchanged.  Shared helpers: tests/gpu_common.py; the `dev` / `oracle` /
`knobs` fixtures: tests/conftest.py.)"""
import ctypes as C  # noqa: F401
import json  # noqa: F401
import os  # noqa: F401
import subprocess  # noqa: F401
import sys  # noqa: F401

import numpy as np  # noqa: F401
import pytest

from tests.conftest import ROOT, scene_path  # noqa: F401
from tests import golden_cases as gc  # noqa: F401
from tests.gpu_common import *  # noqa: F401,F403
from tests.gpu_common import _bits, _counters, _directions, _edge_rays, _generated_assets, _lamp, _mesh_rays, _prim_array, _random_rays, _render_tiles, _ulp_neighbours, _within_bar  # noqa: F401

pytestmark = pytest.mark.gpu


def test_device_math_bit_exact(dev, oracle):
    rng = np.random.default_rng(1)
    x = (rng.random(4096) * 2 * np.pi).astype(np.float32)
    assert np.array_equal(dev.selftest_math(0, x), oracle.math1("sin", x))
    assert np.array_equal(dev.selftest_math(1, x), oracle.math1("cos", x))
    u = (rng.random(4096) * 2 - 1).astype(np.float32)
    assert np.array_equal(dev.selftest_math(2, u), oracle.math1("acos", u))
    a, b = (rng.random(4096) * 4 - 2).astype(np.float32), (rng.random(4096) * 4 - 2).astype(np.float32)
    assert np.array_equal(dev.selftest_math(3, a, b), oracle.math2("atan2", a, b))
    base, ex = rng.random(4096).astype(np.float32), (rng.random(4096) * 60).astype(np.float32)
    assert np.array_equal(dev.selftest_math(4, base, ex), oracle.math2("pow", base, ex))
    e = (rng.random(4096) * 30 - 25).astype(np.float32)
    assert np.array_equal(dev.selftest_math(5, e), oracle.math1("exp", e))
    p = (rng.random(4096) * 3000).astype(np.float32)
    for k in (150.0, 30.0, 300.0, 1.0):
        kk = np.full_like(p, k)
        assert np.array_equal(dev.selftest_math(6, p, kk), np.fmod(p, kk))
    # IEEE division and sqrt on the device (the restatement relies on them)
    num, den = (rng.standard_normal(4096) * 100).astype(np.float32), (rng.standard_normal(4096) * 3).astype(np.float32)
    assert np.array_equal(dev.selftest_math(7, num, den), num / den)
    assert np.array_equal(dev.selftest_math(8, p), np.sqrt(p))


def test_device_checker_matches_oracle(dev, oracle):
    """lambert.rs:58-90 on the device (two shared exact remainders per coordinate) against the oracle's six fmods:
    random points, exact multiples of the periods, one-ulp neighbours of every band edge, both signs, zeros,
    and magnitudes beyond 2^24 (the general path)."""
    import ctypes as C
    rng = np.random.default_rng(5)
    edges = np.array([0, 1, 2, 30, 31, 32, 150, 151, 152, 300, 450, 600, 4500, 16777215, 16777216, 3.0e8], dtype=np.float32)
    near = np.concatenate([edges, np.nextafter(edges, np.float32(np.inf)), np.nextafter(edges, np.float32(-np.inf))])
    pts = np.concatenate([near, -near, (rng.random(3000) * 2000 - 1000).astype(np.float32), np.array([-0.0], dtype=np.float32),
                          (rng.integers(-40, 40, 400) * 30).astype(np.float32), (rng.random(200) * 6e7 - 3e7).astype(np.float32)])
    u = rng.permutation(pts).astype(np.float32)
    v = rng.permutation(pts).astype(np.float32)
    got = dev.selftest_math(9, u, v)
    out = (C.c_float * 3)()
    want = np.empty_like(got)
    for i, (a, b) in enumerate(zip(u, v)):
        oracle.lib().lr_oracle_checker(float(a), float(b), out)
        want[i] = out[0]
    assert np.array_equal(got, want)
    assert set(np.unique(got)) == {np.float32(0.5), np.float32(0.6), np.float32(0.8), np.float32(1.0)}


def test_device_rng_bit_exact(dev, oracle):
    rng = np.random.default_rng(2)
    px = rng.integers(0, 2 ** 22, 512, dtype=np.uint32)
    sm = rng.integers(0, 8192, 512, dtype=np.uint32)
    bk = rng.integers(0, 140, 512, dtype=np.uint32)
    got = dev.selftest_rng(12345, px, sm, bk)
    want = np.stack([oracle.rng_block(12345, int(a), int(b), int(c)) for a, b, c in zip(px, sm, bk)])
    assert np.array_equal(got, want)
    assert got.min() >= 0.0 and got.max() < 1.0


def test_fast_reciprocal_is_ieee_exact(dev):
    """The five-instruction 1/det of the triangle test (lr_math.h rcp_exact_mid) against the compiler's IEEE quotient
    on EVERY float with a biased exponent in 1..252: 4.2e9 bit patterns, all the device code can ever feed it
    (lr_scene_create refuses triangles with |e1| |e2| >= 2^120)."""
    bad2, _bad3, example = dev.selftest_rcp(1, 252)
    assert bad2 == 0, f"reciprocal differs from IEEE for bits {example[0]:#x}"
    # outside that range the short form is NOT the IEEE quotient, which is why the range is enforced
    assert dev.selftest_rcp(0, 0)[0] > 0 and dev.selftest_rcp(253, 254)[0] > 0


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


def test_ibl_map_is_stored_as_rgbe_and_decodes_to_the_same_bits(dev, oracle, knobs):
    """A map that was loaded from an .hdr file holds Radiance values c * 2^(e - 136) (the `image` crate's decode behind
    sky.rs:45-48); lr_scene_create re-encodes every texel, checks the device's decode of the whole map against the caller's
    floats and then keeps 4 B per texel instead of 16.  Same texel (sky.rs:57-78), same f32 bits: lookups equal the oracle's
    and the float4 build's bit for bit, films and counters of the two storage forms are identical."""
    monkeypatch = knobs                                  # (the knob build of the library: the product one reads no LR_* switch, csrc/lr_knobs.h)
    if not _generated_assets():
        pytest.skip("generated assets missing")
    desc = load("ibl-lens.toml", 64, 48)
    rgbe = dev.Scene(desc)
    assert rgbe.sky_texel_bytes() == 4
    monkeypatch.setenv("LR_SKY_FLOAT4", "1")
    f4 = dev.Scene(desc)
    monkeypatch.delenv("LR_SKY_FLOAT4")
    assert f4.sky_texel_bytes() == 16
    d = _directions(np.random.default_rng(41), 300_000)
    a, b, want = rgbe.sky(d), f4.sky(d), oracle.sky_batch(desc, d)
    assert np.array_equal(a.view(np.uint32), want.view(np.uint32)) and np.array_equal(b.view(np.uint32), want.view(np.uint32))
    assert a.max() > 100.0 and len(np.unique(a[:, 0])) > 1000                  # bright texels and the gradient, not a constant
    p = desc.render_params(spp=16, seed=5)
    fa = rgbe.render(p); sa = rgbe.stats()
    fb = f4.render(p); sb = f4.stats()
    assert np.array_equal(fa.view(np.uint32), fb.view(np.uint32))
    assert (sa.samples, sa.segments, sa.shadow_rays, sa.sky_fetches) == (sb.samples, sb.segments, sb.shadow_rays, sb.sky_fetches)
    assert sa.sky_fetches > 0
    rgbe.close(); f4.close()


def test_a_map_that_is_not_rgbe_stays_float4(dev, oracle):
    """The C ABI takes any f32 map (LrSceneDesc.sky.texels).  One texel that is not c * 2^(e - 136) with 8-bit mantissas and
    a shared exponent -- here: every texel scaled by 1.1, and one made negative -- keeps the whole map as float4; lookups
    still equal the oracle's on the same description."""
    if not _generated_assets():
        pytest.skip("generated assets missing")
    desc = load("ibl-lens.toml", 16, 16)
    sky = desc.desc.sky
    n = int(sky.height) * int(sky.height) * 2 * 3
    orig = np.ctypeslib.as_array(sky.texels, shape=(n,)).copy()
    d = _directions(np.random.default_rng(43), 50_000)
    for edit in ("scaled", "negative", "denormal"):
        tex = orig.copy()
        if edit == "scaled":
            tex *= np.float32(1.1)
        elif edit == "negative":
            tex[3 * 12345] = np.float32(-1.0)
        else:
            tex[3 * 777 + 1] = np.float32(1e-40)                               # representable only as a denormal: e < 10
        keep = tex                                                             # the description reads the array while the scenes are created
        desc.desc_ptr.contents.sky.texels = keep.ctypes.data_as(C.POINTER(C.c_float))
        sc = dev.Scene(desc)
        assert sc.sky_texel_bytes() == 16, edit
        assert np.array_equal(sc.sky(d).view(np.uint32), oracle.sky_batch(desc, d).view(np.uint32)), edit
        sc.close()
    desc.desc_ptr.contents.sky.texels = orig.ctypes.data_as(C.POINTER(C.c_float))
    sc = dev.Scene(desc)
    assert sc.sky_texel_bytes() == 4
    sc.close()


def test_golden_function_vectors(dev):
    """Per-function vectors for SURVEY 8(a) rows a8-a19 (tests/golden/functions.npz) against the device functions the render kernels
    call, through the diagnostic entry points of include/lumilly_hip_diag.h.  Everything that decides something -- which primitive
    and at what distance, which emitter and which point on it, sampled directions, camera rays, which texel -- must be the SAME
    BITS; BSDF values, pdfs and the thin-lens weight (radiance-only, 1-ulp reciprocals on the device: lr_kernels.h rcp_r) to 1e-5."""
    from tests import golden_cases as gc
    fn = np.load(gc.FUNCTIONS)
    # a8 / a10 / a11: closest hit, flat loop (12 triangles + 2 spheres) and the 4-wide tree over 100k triangles
    desc = gc.load_scene("cbox-spheres.toml", None, 16, 16)
    scene = dev.Scene(desc)
    o, d = gc.rays_in_box(512, (0, 0, -100), (556, 548, 560), 21)
    prim, t = scene.intersect(o, d)
    assert np.array_equal(prim, fn["cbox_prim"]) and np.array_equal(_bits(np.where(prim >= 0, t, 0)), _bits(np.where(fn["cbox_prim"] >= 0, fn["cbox_t"], 0)))
    # a12: emitter pick and the sampled point (objects.rs:37-51, triangle.rs:140-149)
    xi = np.random.default_rng(23).random((64, 4), dtype=np.float32)
    assert np.array_equal(scene.emitter_pick(xi[:, 1]), fn["emit_pick"])
    es = scene.emission_sample(xi)
    assert np.array_equal(_bits(es[:, :3]), _bits(fn["emit_sample"][:, :3]))
    assert np.allclose(es[:, 3], fn["emit_sample"][:, 3], rtol=1e-6, atol=0)
    scene.close()
    # a13-a16, f1: the five BSDFs
    inp = gc.material_inputs()
    for name in gc.MATERIALS:
        got = dev.selftest_material(gc.material(name), inp)
        ref = fn["bsdf_" + name]
        assert np.array_equal(_bits(got[:, :3]), _bits(ref[:, :3])), name                  # the sampled direction decides the next ray
        both = np.isfinite(ref[:, 3:]) & np.isfinite(got[:, 3:])
        assert np.array_equal(np.isfinite(ref[:, 3:]), np.isfinite(got[:, 3:])), name
        assert np.allclose(got[:, 3:][both], ref[:, 3:][both], rtol=1e-5, atol=1e-30), name
    # a2 / a3 / f3: cameras
    for cam, (sc_name, edit, gen) in gc.CAMERA_SCENES.items():
        if gen and not gc.have_generated_assets():
            continue
        desc = gc.load_scene(sc_name, edit, 64, 48)
        scene = dev.Scene(desc)
        xy, xi4 = gc.camera_inputs(64, 48)
        got, ref = scene.camera_samples(xy, xi4), fn["camera_" + cam]
        assert np.array_equal(_bits(got[:, :6]), _bits(ref[:, :6])), cam                   # origin and direction
        assert np.allclose(got[:, 6], ref[:, 7], rtol=1e-6), cam                           # geometry term (out8[7] of the oracle hook)
        scene.close()
    # a18: math spec
    for name, (fid, unary) in gc.MATH_CASES.items():
        a, b = gc.math_inputs(name)
        got = dev.selftest_math(fid, a, b)
        ref = fn["math_" + name]
        assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.array_equal(_bits(got)[~np.isnan(ref)], _bits(ref)[~np.isnan(ref)]), name
    if gc.have_generated_assets():
        desc = gc.load_scene("mesh-box.toml", None, 16, 16)
        scene = dev.Scene(desc)
        o, d = gc.rays_at(**gc.MESH_RAYS)
        prim, t = scene.intersect(o, d)
        assert np.array_equal(prim, fn["mesh_prim"]) and np.array_equal(_bits(np.where(prim >= 0, t, 0)), _bits(np.where(fn["mesh_prim"] >= 0, fn["mesh_t"], 0)))
        scene.close()
        # a19: IBL texel at the poles, on the axes and across the u seam
        desc = gc.load_scene("ibl-lens.toml", None, 16, 16)
        scene = dev.Scene(desc)
        assert np.array_equal(_bits(scene.sky(gc.sky_directions())), _bits(fn["sky_rgb"]))
        scene.close()


def test_ggx_at_grazing_incidence_against_the_oracle(dev, oracle):
    """ggx.rs:27-47's G, D and Fresnel terms (and the thin-lens weight) use the 1-ulp hardware reciprocal / square root on the device
    (lr_kernels.h rcp_r / sqrt_r: radiance-only values; no decision reads them) -- admitted difference of DESIGN.md section 2.  Where the
    exact form runs out of range first is grazing incidence: c = out_ . n down to 1e-20, so that c * c is denormal or zero and
    v_rcp_f32 flushes it where the IEEE division would not.  Sampled directions stay the SAME BITS; values and pdfs stay within 1e-5
    relative (or 1e-30 absolute: both sides are at the bottom of the float range there), and what is non-finite is non-finite on
    both sides."""
    import ctypes as C
    from tests import golden_cases as gc
    rng = np.random.default_rng(51)
    n = 512
    nrm = np.tile(np.array([[0.0, 1.0, 0.0]], dtype=np.float32), (n, 1))
    cosv = (10.0 ** rng.uniform(-20, -1, n)).astype(np.float32)
    phi = rng.uniform(0, 2 * np.pi, n)
    out_ = np.stack([np.cos(phi) * np.sqrt(1 - cosv.astype(np.float64) ** 2), cosv, np.sin(phi) * np.sqrt(1 - cosv.astype(np.float64) ** 2)], axis=1).astype(np.float32)
    inp = np.concatenate([out_, nrm, rng.uniform(-100, 100, (n, 3)).astype(np.float32), rng.random((n, 3), dtype=np.float32), np.ones((n, 1), np.float32)], axis=1).astype(np.float32)
    f3 = lambda v: (C.c_float * len(v))(*[float(x) for x in v])
    for rough in (0.8, 0.2, 0.05):
        m = gc.material("ggx"); m.param[0] = rough
        got = dev.selftest_material(m, inp)
        want = np.zeros_like(got)
        L = oracle.lib()
        for i, a in enumerate(inp):
            in3, pdf, rgb, coef = (C.c_float * 3)(), C.c_float(), (C.c_float * 3)(), (C.c_float * 3)()
            L.lr_oracle_material_sample(C.byref(m), f3(a[0:3]), f3(a[3:6]), f3(a[9:12]), in3, C.byref(pdf))
            L.lr_oracle_material_brdf(C.byref(m), f3(a[0:3]), in3, f3(a[3:6]), f3(a[6:9]), rgb)
            L.lr_oracle_material_coef(C.byref(m), f3(a[0:3]), f3(a[3:6]), float(a[12]), coef)
            want[i] = list(in3) + [pdf.value] + list(rgb) + list(coef)
        assert np.array_equal(got[:, :3].view(np.uint32), want[:, :3].view(np.uint32)), rough       # the next ray: exact
        gv, wv = got[:, 3:7].astype(np.float64), want[:, 3:7].astype(np.float64)                    # pdf, brdf rgb
        fin = np.isfinite(wv)
        # v_rcp_f32 flushes a denormal c * c to zero: the device may reach inf / 0 where the exact form is still ~1e38 / ~1e-38
        edge = ~np.isfinite(gv) | ~fin | (np.abs(wv) > 1e30) | (np.abs(wv) < 1e-30)
        ok = np.isclose(gv, wv, rtol=1e-5, atol=1e-30) | edge
        assert ok.all(), (rough, np.argwhere(~ok)[:4], gv[~ok][:4], wv[~ok][:4])
        # ... and what such a vertex contributes is the same to the film's tolerance: brdf * cos / pdf, as scene.rs:96-99 forms it
        cosi = np.abs(got[:, 1].astype(np.float64))
        with np.errstate(all="ignore"):
            cg, cw = gv[:, 1] * cosi / gv[:, 0], wv[:, 1] * cosi / wv[:, 0]
        both = np.isfinite(cg) & np.isfinite(cw)
        assert both.mean() > 0.9 and np.allclose(cg[both], cw[both], rtol=1e-4, atol=1e-6), rough
