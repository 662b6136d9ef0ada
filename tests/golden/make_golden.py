"""Writes tests/golden/ with the CPU oracle: film crops (*.npy) and the per-function vectors (functions.npz).

The reference cannot run here (no Rust toolchain, unseeded RNG), so these vectors are the oracle's own output at fixed
seeds: they pin the oracle against drift and give the GPU tests a committed target that does not need the oracle.
What is generated is defined in tests/golden_cases.py.  Run from the repo root (after __graft_entry__.build(), which
generates the mesh and the HDR map):   python tests/golden/make_golden.py
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from oracle import binding as oracle  # noqa: E402
from tests import golden_cases as gc  # noqa: E402


def f3(v):
    return (C.c_float * len(v))(*[float(x) for x in v])


def films():
    for case in gc.FILM_CASES:
        name, edit, w, h, spp, integ, seed, gen = case
        if gen and not gc.have_generated_assets():
            print("skipped (generated assets missing):", gc.film_name(case)); continue
        d = gc.load_scene(name, edit, w, h)
        img = oracle.render(d, d.render_params(spp=spp, seed=seed, integrator=integ), mode=oracle.OWNBOX)
        path = os.path.join(gc.GOLDEN, gc.film_name(case))
        np.save(path, img)
        print(path, img.shape, float(np.nanmean(img)))


def stated_spp_films():
    for case in gc.STATED_SPP_CASES:
        name, edit, w, h, spp, integ, seed, gen = case
        if gen and not gc.have_generated_assets():
            print("skipped (generated assets missing):", gc.film_name(case)); continue
        d = gc.load_scene(name, edit, w, h)
        img = oracle.render(d, d.render_params(spp=spp, seed=seed, integrator=integ), mode=oracle.BVH, pad=0.0)
        path = os.path.join(gc.GOLDEN, gc.film_name(case))
        np.save(path, img)
        print(path, img.shape, float(np.nanmean(img)), float(np.nanmax(img)))


def stated_size_films():
    """configs[0..4] (+ the Phong / Blinn-Phong rows) at their STATED film size and spp: the pixels of golden_cases.stated_tiles,
    rendered through the oracle's reference-literal mode (SAH tree, collect-all-candidates walk, pad 0), packed in tile order."""
    for key, (name, edit, w, h, spp, integ, seed, gen, rows) in gc.STATED_SIZE_CASES.items():
        if gen and not gc.have_generated_assets():
            print("skipped (generated assets missing):", gc.stated_name(key)); continue
        d = gc.load_scene(name, edit, w, h)
        tl = gc.stated_tiles(w, h, rows)
        img, st = oracle.render(d, d.render_params(spp=spp, seed=seed, integrator=integ), gc.tile_array(tl), len(tl), mode=oracle.BVH, pad=0.0,
                                with_stats=True, fast=True)
        packed = gc.pack_tiles(img, tl)
        path = os.path.join(gc.GOLDEN, gc.stated_name(key))
        np.save(path, packed)
        print(path, packed.shape, float(np.nanmean(packed)), float(np.nanmax(packed)), f"{st.seconds:.1f} s")


def functions():
    out = {}
    L = oracle.lib()
    # a9  AABB::is_intersect (aabb.rs:74-92)
    box, o, d = gc.aabb_inputs()
    out["aabb_hit"] = np.array([L.lr_oracle_aabb_is_intersect(f3(box[i]), f3(o[i]), f3(d[i])) for i in range(len(box))], dtype=np.uint8)
    # a10 / a11 / a8: closest hit over the flat Cornell scene (12 triangles + 2 spheres) and over the 100k-triangle tree
    desc = gc.load_scene("cbox-spheres.toml", None, 16, 16)
    o, d = gc.rays_in_box(512, (0, 0, -100), (556, 548, 560), 21)
    prim, t = oracle.intersect(desc, o, d, mode=oracle.OWNBOX)
    out["cbox_prim"], out["cbox_t"] = prim, t
    if gc.have_generated_assets():
        desc = gc.load_scene("mesh-box.toml", None, 16, 16)
        o, d = gc.rays_at(**gc.MESH_RAYS)
        prim, t = oracle.intersect(desc, o, d, mode=oracle.OWNBOX)
        out["mesh_prim"], out["mesh_t"] = prim, t
    # a13-a16 (+ f1): sample / brdf / coef of the five BSDFs
    inp = gc.material_inputs()
    for name in gc.MATERIALS:
        m = gc.material(name)
        res = np.zeros((len(inp), 10), dtype=np.float32)
        for i, a in enumerate(inp):
            in3, pdf, rgb, coef = (C.c_float * 3)(), C.c_float(), (C.c_float * 3)(), (C.c_float * 3)()
            L.lr_oracle_material_sample(C.byref(m), f3(a[0:3]), f3(a[3:6]), f3(a[9:12]), in3, C.byref(pdf))
            L.lr_oracle_material_brdf(C.byref(m), f3(a[0:3]), in3, f3(a[3:6]), f3(a[6:9]), rgb)
            L.lr_oracle_material_coef(C.byref(m), f3(a[0:3]), f3(a[3:6]), float(a[12]), coef)
            res[i] = list(in3) + [pdf.value] + list(rgb) + list(coef)
        out["bsdf_" + name] = res
    # a12: emitter pick + sample (objects.rs:37-51, triangle.rs:140-149)
    desc = gc.load_scene("cbox-spheres.toml", None, 16, 16)
    xi = np.random.default_rng(23).random((64, 4), dtype=np.float32)
    out["emit_pick"] = oracle.emitter_pick(desc, xi[:, 1])[0]
    out["emit_sample"] = oracle.emission_sample(desc, xi)
    # a2 / a3 / f3: the three cameras
    for cam, (scene, edit, gen) in gc.CAMERA_SCENES.items():
        if gen and not gc.have_generated_assets():
            continue
        desc = gc.load_scene(scene, edit, 64, 48)
        xy, xi4 = gc.camera_inputs(64, 48)
        res = np.zeros((len(xy), 8), dtype=np.float32)
        for i in range(len(xy)):
            o8 = (C.c_float * 8)()
            L.lr_oracle_camera_sample(C.byref(desc.desc.camera), int(xy[i, 0]), int(xy[i, 1]), f3(xi4[i]), o8)
            res[i] = list(o8)
        out["camera_" + cam] = res
    # a19: IBL texel at the poles, the axes and the u seam
    if gc.have_generated_assets():
        desc = gc.load_scene("ibl-lens.toml", None, 16, 16)
        out["sky_rgb"] = oracle.sky_batch(desc, gc.sky_directions())
    # a18: the deterministic math spec
    for name in gc.MATH_CASES:
        a, b = gc.math_inputs(name)
        out["math_" + name] = oracle.math_batch(name, a, b)
    np.savez(gc.FUNCTIONS, **out)
    print(gc.FUNCTIONS, {k: v.shape for k, v in out.items()})


def denormal_pdf_pixels():
    """golden_cases.DENORMAL_PDF_PIXELS: the oracle's value of each pixel after samples 0 .. k of the stated Phong row (finite: scene.rs:101
    divides a denormal BRDF value by a denormal pdf)."""
    name, edit, w, h, spp, integ, seed, gen, rows = gc.STATED_SIZE_CASES["c3p"]
    d = gc.load_scene(name, edit, w, h)
    out = np.zeros((len(gc.DENORMAL_PDF_PIXELS), 3), dtype=np.float32)
    for i, (x, y, k) in enumerate(gc.DENORMAL_PDF_PIXELS):
        out[i] = oracle.render(d, d.render_params(spp=k + 1, seed=seed, integrator=integ), gc.one_pixel_tile(x, y), 1, mode=oracle.BVH, pad=0.0)[y, x]
    path = os.path.join(gc.GOLDEN, gc.DENORMAL_PDF_FIXTURE)
    np.save(path, out)
    print(path, out)


if __name__ == "__main__":
    what = sys.argv[1:] or ["films", "stated_spp", "stated_size", "functions", "denormal_pdf"]
    if "denormal_pdf" in what:
        denormal_pdf_pixels()
    if "films" in what:
        films()
    if "stated_spp" in what:
        stated_spp_films()
    if "stated_size" in what:
        stated_size_films()
    if "functions" in what:
        functions()
