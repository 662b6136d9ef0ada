"""CPU side of the golden fixtures (tests/golden/, defined in tests/golden_cases.py): the oracle must still reproduce every
committed vector bit for bit -- they pin it against silent drift (nothing can be captured from the reference: SURVEY 8c) -- and
the reference-literal pieces that have no device counterpart (AABB::is_intersect, aabb.rs:74-92) are checked here only."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import binding as oracle
from tests import golden_cases as gc


def f3(v):
    return (C.c_float * len(v))(*[float(x) for x in v])


@pytest.fixture(scope="module")
def fn():
    return np.load(gc.FUNCTIONS)


@pytest.mark.parametrize("case", gc.FILM_CASES, ids=[gc.film_name(c).replace(".npy", "") for c in gc.FILM_CASES])
def test_oracle_reproduces_the_film_crops(case):
    name, edit, w, h, spp, integ, seed, gen = case
    if gen and not gc.have_generated_assets():
        pytest.skip("generated assets missing (run __graft_entry__.build())")
    d = gc.load_scene(name, edit, w, h)
    img = oracle.render(d, d.render_params(spp=spp, seed=seed, integrator=integ), mode=oracle.OWNBOX)
    ref = np.load(os.path.join(gc.GOLDEN, gc.film_name(case)))
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))          # bit for bit, NaNs included
    assert np.isfinite(ref).mean() > 0.99 and float(np.nanmax(ref)) > 0.0


@pytest.mark.parametrize("case", gc.STATED_SPP_CASES, ids=[gc.film_name(c).replace(".npy", "") for c in gc.STATED_SPP_CASES])
def test_oracle_reproduces_the_stated_spp_crops(case):
    """The BASELINE configs at their stated spp (1024 / 4096 / 2048 / 8192) on an 8 x 6 film, through the reference-literal tree."""
    name, edit, w, h, spp, integ, seed, gen = case
    if gen and not gc.have_generated_assets():
        pytest.skip("generated assets missing (run __graft_entry__.build())")
    d = gc.load_scene(name, edit, w, h)
    img, st = oracle.render(d, d.render_params(spp=spp, seed=seed, integrator=integ), mode=oracle.BVH, pad=0.0, with_stats=True)
    ref = np.load(os.path.join(gc.GOLDEN, gc.film_name(case)))
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    assert st.samples == w * h * spp and np.isfinite(ref).all() and float(ref.max()) > 0.0


def test_aabb_is_intersect_vectors(fn):
    """aabb.rs:74-92, literal (three divisions, early exit): the device replaces it by conservative quantised boxes that only prune,
    so this vector pins the oracle's reference-literal traversal mode only."""
    box, o, d = gc.aabb_inputs()
    got = np.array([oracle.lib().lr_oracle_aabb_is_intersect(f3(box[i]), f3(o[i]), f3(d[i])) for i in range(len(box))], dtype=np.uint8)
    assert np.array_equal(got, fn["aabb_hit"])
    assert 8 < int(got.sum()) < 56                                         # both outcomes are exercised
    # independent statement of the slab test in f64 on the rays that are not axis-parallel: same verdict away from grazing contact
    for i in range(len(box)):
        if np.any(d[i] == 0):
            continue
        t0 = (box[i, :3].astype(np.float64) - o[i]) / d[i]; t1 = (box[i, 3:].astype(np.float64) - o[i]) / d[i]
        tn, tf = np.minimum(t0, t1).max(), np.maximum(t0, t1).min()
        if abs(tn - tf) > 1e-3:
            assert bool(got[i]) == bool(tn <= tf), i


def test_oracle_reproduces_the_function_vectors(fn):
    L = oracle.lib()
    desc = gc.load_scene("cbox-spheres.toml", None, 16, 16)
    o, d = gc.rays_in_box(512, (0, 0, -100), (556, 548, 560), 21)
    prim, t = oracle.intersect(desc, o, d, mode=oracle.OWNBOX)
    assert np.array_equal(prim, fn["cbox_prim"]) and np.array_equal(t.view(np.uint32), fn["cbox_t"].view(np.uint32))
    assert (prim >= 12).any() and (prim < 12).any() and (prim < 0).any() or (prim >= 0).all()     # spheres and triangles are hit
    # the tree (reference-literal BVH mode, bvh.rs:131-141) returns what the definition returns
    prim2, t2 = oracle.intersect(desc, o, d, mode=oracle.BVH)
    assert np.array_equal(prim2, prim) and np.array_equal(t2.view(np.uint32), t.view(np.uint32))
    inp = gc.material_inputs()
    for name in gc.MATERIALS:
        m = gc.material(name)
        ref = fn["bsdf_" + name]
        for i, a in enumerate(inp):
            in3, pdf, rgb, coef = (C.c_float * 3)(), C.c_float(), (C.c_float * 3)(), (C.c_float * 3)()
            L.lr_oracle_material_sample(C.byref(m), f3(a[0:3]), f3(a[3:6]), f3(a[9:12]), in3, C.byref(pdf))
            L.lr_oracle_material_brdf(C.byref(m), f3(a[0:3]), in3, f3(a[3:6]), f3(a[6:9]), rgb)
            L.lr_oracle_material_coef(C.byref(m), f3(a[0:3]), f3(a[3:6]), float(a[12]), coef)
            got = np.array(list(in3) + [pdf.value] + list(rgb) + list(coef), dtype=np.float32)
            assert np.array_equal(got.view(np.uint32), ref[i].view(np.uint32)), (name, i)
        assert np.allclose(np.linalg.norm(ref[:, :3], axis=1), 1.0, atol=1e-5), name          # sampled directions are unit vectors
    xi = np.random.default_rng(23).random((64, 4), dtype=np.float32)
    assert np.array_equal(oracle.emitter_pick(desc, xi[:, 1])[0], fn["emit_pick"])
    assert np.array_equal(oracle.emission_sample(desc, xi).view(np.uint32), fn["emit_sample"].view(np.uint32))
    assert set(fn["emit_pick"].tolist()) == {0, 1}
    for name in gc.MATH_CASES:
        a, b = gc.math_inputs(name)
        assert np.array_equal(oracle.math_batch(name, a, b).view(np.uint32), fn["math_" + name].view(np.uint32)), name


@pytest.mark.parametrize("cam", sorted(gc.CAMERA_SCENES))
def test_oracle_reproduces_the_camera_vectors(fn, cam):
    scene, edit, gen = gc.CAMERA_SCENES[cam]
    if gen and not gc.have_generated_assets():
        pytest.skip("generated assets missing")
    desc = gc.load_scene(scene, edit, 64, 48)
    xy, xi4 = gc.camera_inputs(64, 48)
    ref = fn["camera_" + cam]
    for i in range(len(xy)):
        o8 = (C.c_float * 8)()
        oracle.lib().lr_oracle_camera_sample(C.byref(desc.desc.camera), int(xy[i, 0]), int(xy[i, 1]), f3(xi4[i]), o8)
        assert np.array_equal(np.array(list(o8), dtype=np.float32).view(np.uint32), ref[i].view(np.uint32)), (cam, i)
    assert np.allclose(np.linalg.norm(ref[:, 3:6], axis=1), 1.0, atol=1e-5)


def test_oracle_reproduces_the_sky_and_mesh_vectors(fn):
    if not gc.have_generated_assets():
        pytest.skip("generated assets missing")
    desc = gc.load_scene("ibl-lens.toml", None, 16, 16)
    assert np.array_equal(oracle.sky_batch(desc, gc.sky_directions()).view(np.uint32), fn["sky_rgb"].view(np.uint32))
    desc = gc.load_scene("mesh-box.toml", None, 16, 16)
    o, d = gc.rays_at(**gc.MESH_RAYS)
    prim, t = oracle.intersect(desc, o, d, mode=oracle.OWNBOX)
    assert np.array_equal(prim, fn["mesh_prim"]) and np.array_equal(t.view(np.uint32), fn["mesh_t"].view(np.uint32))
    assert (prim >= 0).mean() > 0.9 and len(np.unique(prim)) > 100


@pytest.mark.parametrize("key", ["c1"] if os.environ.get("LUMILLY_TEST_LIGHT") else ["c1", "c2", "c3p"])      # (light: the sanitizer run)
def test_oracle_reproduces_the_stated_size_tiles(key):
    """BASELINE configs at their stated FILM SIZE and spp (round 6): 64 scattered 16-px tiles of the full-size film (+ the rows through
    the box's edges for configs[1]), through the reference-literal walk.  (c3, c3b, c4, c5 take 10-50 s each on 8 cores: the GPU suite
    checks the device against all seven fixtures, and against the live oracle, tests/test_gpu_film.py.)"""
    name, edit, w, h, spp, integ, seed, gen, rows = gc.STATED_SIZE_CASES[key]
    d = gc.load_scene(name, edit, w, h)
    tl = gc.stated_tiles(w, h, rows)
    img, st = oracle.render(d, d.render_params(spp=spp, seed=seed, integrator=integ), gc.tile_array(tl), len(tl), mode=oracle.BVH, pad=0.0,
                            with_stats=True, fast=True)
    ref = np.load(os.path.join(gc.GOLDEN, gc.stated_name(key)))
    got = gc.pack_tiles(img, tl)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    assert st.samples == len(ref) * spp and len(ref) >= 16384


def test_oracle_keeps_a_denormal_pdf_finite():
    """golden_cases.DENORMAL_PDF_PIXELS: the oracle reproduces the fixture bit for bit, and every value is finite (the pdf is denormal, not 0)."""
    name, edit, w, h, spp, integ, seed, gen, rows = gc.STATED_SIZE_CASES["c3p"]
    d = gc.load_scene(name, edit, w, h)
    ref = np.load(os.path.join(gc.GOLDEN, gc.DENORMAL_PDF_FIXTURE))
    assert ref.shape == (len(gc.DENORMAL_PDF_PIXELS), 3) and np.isfinite(ref).all() and (ref > 0).all()
    for i, (x, y, k) in enumerate(gc.DENORMAL_PDF_PIXELS):
        got = oracle.render(d, d.render_params(spp=k + 1, seed=seed, integrator=integ), gc.one_pixel_tile(x, y), 1, mode=oracle.BVH, pad=0.0)[y, x]
        assert np.array_equal(got.view(np.uint32), ref[i].view(np.uint32)), (x, y, k, got, ref[i])


def test_stated_tiles_are_disjoint_and_inside_the_film():
    for key, (name, edit, w, h, spp, integ, seed, gen, rows) in gc.STATED_SIZE_CASES.items():
        tl = gc.stated_tiles(w, h, rows)
        m = gc.tile_mask(w, h, tl)                       # (asserts disjointness)
        assert int(m.sum()) >= 16384 and all(x >= 0 and y >= 0 and x + tw <= w and y + th <= h for x, y, tw, th in tl), key
        for r in rows:
            assert m[r].all(), (key, r)
