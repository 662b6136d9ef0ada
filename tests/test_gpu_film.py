"""GPU parity -- film.

Films: lr_render against the oracle on identical scene + seed (per-channel L-infinity < 1e-4 on the linear film, a pixel brighter than 1 gets 1e-4 of its own
value; identical sample / segment / shadow-ray / sky-fetch counters), against the committed golden fixtures without the oracle, at the stated spp and at the stated
FILM SIZE of every BASELINE config, and the properties a film must have whatever the tiling, slot count, banding and chunk schedule.

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


@pytest.mark.parametrize("name,w,h,spp,integ", CASES)
def test_image_parity(dev, oracle, name, w, h, spp, integ):
    desc = load(name, w, h)
    params = desc.render_params(spp=spp, seed=11, integrator=integ)
    scene = dev.Scene(desc)
    img = scene.render(params)
    ref, ost = oracle.render(desc, params, with_stats=True)
    st = scene.stats()
    # identical RNG streams => identical path statistics, an exact check of every discrete decision
    assert st.samples == w * h * spp == ost.samples
    assert st.segments == ost.segments
    assert st.shadow_rays == ost.shadow_rays
    assert linf(img, ref) < TOL
    assert ref.max() > 0.05
    scene.close()


@pytest.mark.parametrize("mat", ["phong", "blinn-phong"])
def test_image_parity_phong_family(dev, oracle, mat):
    def edit(t):
        out, k = [], 0
        alphas = ["1", "5", "10", "20"]
        lines = t.split("\n")
        i = 0
        while i < len(lines):
            ln = lines[i]
            if ln.strip() == 'type = "ggx"':
                out.append(f'type = "{mat}"')
            elif ln.startswith("roughness"):
                out.append(f"alpha = {alphas[k]}"); k += 1
            elif ln.startswith("ior"):
                pass
            else:
                out.append(ln)
            i += 1
        return "\n".join(out)
    desc = load("brdf-row.toml", 64, 36, text_edit=edit)
    params = desc.render_params(spp=32, seed=5)
    scene = dev.Scene(desc)
    img = scene.render(params)
    ref = oracle.render(desc, params)
    both_nan = np.isnan(img) & np.isnan(ref)          # the reference's powf(negative, a) NaN, reproduced on both sides
    assert np.array_equal(np.isnan(img), np.isnan(ref))
    assert float(np.max(np.abs(np.where(both_nan, 0, img) - np.where(both_nan, 0, ref)))) < TOL
    scene.close()


def test_tiles_and_slots_do_not_change_the_image(dev):
    """RNG is keyed by (seed, pixel, sample); chunking depends on spp only: any tiling, any slot count
    and therefore any GPU count gives the same film bit for bit."""
    from lumillyrender_amd import host
    desc = load("cbox-spheres.toml", 50, 38)          # ragged against 16-pixel tiles
    params = desc.render_params(spp=20, seed=3)
    scene = dev.Scene(desc)
    full = scene.render(params)
    out = np.zeros_like(full)
    for rank in range(3):
        tiles, n = host.tiles(50, 38, 16, rank, 3)
        scene.render(params, tiles, n, out=out)
    assert np.array_equal(full, out)
    p2 = desc.render_params(spp=20, seed=3, path_slots=512)
    assert np.array_equal(full, scene.render(p2))
    scene.close()


def test_edge_cases(dev, oracle):
    from lumillyrender_amd import abi, host
    desc = load("cbox-spheres.toml", 8, 8)
    scene = dev.Scene(desc)
    params = desc.render_params(spp=1, seed=0)
    # empty tile list, zero-area tiles: nothing is written
    canvas = np.full((8, 8, 3), -1.0, dtype=np.float32)
    tiles = (abi.LrTile * 2)()
    tiles[0].x0, tiles[0].y0, tiles[0].w, tiles[0].h = 2, 2, 0, 3
    tiles[1].x0, tiles[1].y0, tiles[1].w, tiles[1].h = 4, 4, 1, 1
    scene.render(params, tiles, 0, out=canvas)
    assert (canvas == -1).all()
    scene.render(params, tiles, 2, out=canvas)
    assert (canvas[4, 4] >= 0).all() and (np.delete(canvas.reshape(-1, 3), 4 * 8 + 4, axis=0) == -1).all()
    # spp = 1 on a single pixel agrees with the oracle
    ref = oracle.render(desc, params)
    assert np.max(np.abs(canvas[4, 4] - ref[4, 4])) < TOL
    # tile outside the film is rejected, not clipped
    tiles[1].x0 = 8
    with pytest.raises(host.LumillyError):
        scene.render(params, tiles, 2, out=canvas)
    scene.close()


def test_row_stride_and_ragged_tiles(dev):
    """lr_render writes rows `row_stride_floats` apart (an Img with padding, or a window of a larger canvas) and takes
    any set of non-overlapping rectangles, not only the 64x64 grid: the assembled window equals a plain render."""
    import ctypes as C
    from lumillyrender_amd import abi
    W, H = 70, 45
    desc = load("cbox-spheres.toml", W, H)
    scene = dev.Scene(desc)
    params = desc.render_params(spp=4, seed=8)
    want = scene.render(params)
    rects = [(0, 0, 33, 20), (33, 0, 37, 7), (33, 7, 37, 13), (0, 20, 70, 1), (0, 21, 1, 24), (1, 21, 69, 24)]   # x0, y0, w, h: a ragged cover
    assert sum(w * h for _, _, w, h in rects) == W * H
    tiles = (abi.LrTile * len(rects))()
    for t, (x0, y0, w, h) in zip(tiles, rects):
        t.x0, t.y0, t.w, t.h = x0, y0, w, h
    stride = W * 3 + 11                                              # floats per row of the destination
    canvas = np.full((H + 2, stride), -7.0, dtype=np.float32)
    rc = dev.lib().lr_render(scene._h, C.byref(params), tiles, len(rects), canvas[1:].ctypes.data_as(C.POINTER(C.c_float)), stride)
    assert rc == 0
    got = canvas[1:1 + H, :W * 3].reshape(H, W, 3)
    assert np.array_equal(got, want)
    assert (canvas[0] == -7).all() and (canvas[-1] == -7).all() and (canvas[1:1 + H, W * 3:] == -7).all()   # nothing outside the window
    # a stride smaller than a row is refused
    assert dev.lib().lr_render(scene._h, C.byref(params), tiles, len(rects), canvas.ctypes.data_as(C.POINTER(C.c_float)), W * 3 - 1) == abi.LR_EINVAL
    scene.close()


@pytest.mark.parametrize("name,integ", [("mesh-box.toml", None), ("ibl-lens.toml", None)])
def test_mesh_scene_parity(dev, oracle, name, integ):
    """C4 / C5 class: 100k-triangle mesh (deep BVH), thin-lens camera, IBL sky, GGX.  The oracle runs its
    padded-tree mode, which returns exactly the brute-force closest hit."""
    if not _generated_assets():
        pytest.skip("generated assets missing (run __graft_entry__.build())")
    desc = load(name, 48, 36)
    params = desc.render_params(spp=8, seed=21, integrator=integ)
    scene = dev.Scene(desc)
    img = scene.render(params)
    ref, ost = oracle.render(desc, params, mode=oracle.OWNBOX_TREE, with_stats=True)
    st = scene.stats()
    assert (st.samples, st.segments, st.shadow_rays) == (ost.samples, ost.segments, ost.shadow_rays)
    if name == "ibl-lens.toml":
        assert st.sky_fetches == ost.sky_fetches and st.sky_fetches > 0
    # the IBL has ~1e3 texels: a pixel brighter than 1 gets 1e-4 of ITS OWN value (per pixel: DESIGN section 2, round 5)
    assert np.all(np.abs(img - ref) < TOL * np.maximum(1.0, np.abs(ref)))
    scene.close()


def test_ideal_refraction_parity(dev, oracle):
    """dielectric with Fresnel roulette and Beer absorption (ideal_refraction.rs)."""
    def edit(t):
        parts = t.rsplit('material = "matte"', 1)
        t = parts[0] + 'material = "glass"' + parts[1]
        return t.replace('  { name = "dark",', GLASS + '  { name = "dark",')
    desc = load("cbox-spheres.toml", 40, 40, text_edit=edit)
    assert any(m["type"] == 4 for m in desc.dump(0)["materials"])
    params = desc.render_params(spp=24, seed=9)
    scene = dev.Scene(desc)
    img = scene.render(params)
    ref, ost = oracle.render(desc, params, with_stats=True)
    st = scene.stats()
    assert (st.segments, st.shadow_rays) == (ost.segments, ost.shadow_rays)
    assert np.array_equal(np.isnan(img), np.isnan(ref))
    ok = ~np.isnan(ref)
    assert np.all(np.abs(img[ok] - ref[ok]) < TOL * np.maximum(1.0, np.abs(ref[ok])))
    scene.close()


def test_omnidirectional_camera_parity(dev, oracle):
    """camera.rs:137-197."""
    def edit(t):
        return t.replace('type = "ideal-pinhole"\nfov = 39.3077\n', 'type = "omnidirectional"\n').replace("[278, 273, -800]", "[278, 273, 100]")
    desc = load("cbox-spheres.toml", 48, 24, text_edit=edit)
    assert desc.desc.camera.type == 2
    params = desc.render_params(spp=16, seed=2)
    scene = dev.Scene(desc)
    img = scene.render(params)
    ref = oracle.render(desc, params)
    assert linf(img, ref) < TOL
    scene.close()


def test_full_size_properties(dev):
    """BASELINE-size film (1024x1024) where the oracle is too slow: size-independent properties.
      * sample count and per-path statistics are exact;
      * doubling every emitter doubles the film exactly (scaling by 2 commutes with every rounding);
      * a tiled render equals the untiled one bit for bit."""
    from lumillyrender_amd import host
    W = H = 1024
    desc = load("cbox-spheres.toml", W, H)
    params = desc.render_params(spp=8, seed=5)
    scene = dev.Scene(desc)
    a = scene.render(params)
    st = scene.stats()
    assert st.samples == W * H * 8 and st.segments > st.samples and st.shadow_rays > 0
    assert np.isfinite(a).all() and a.min() >= 0
    out = np.zeros_like(a)
    for rank in range(4):
        tiles, n = host.tiles(W, H, 64, rank, 4)
        scene.render(params, tiles, n, out=out)
    assert np.array_equal(a, out)
    scene.close()
    d2 = load("cbox-spheres.toml", W, H, text_edit=lambda t: t.replace("intensity = 0.7", "intensity = 1.4"))
    s2 = dev.Scene(d2)
    b = s2.render(params)
    assert np.array_equal(b, a * np.float32(2.0))
    s2.close()


def test_device_film_output_stage(dev, tmp_path):
    """quantisation on the device (main.rs:171-173 gamma + truncation; img.rs:40-50 RGBE)
    against the host writers, which are pinned on CPU by tests/test_host_loader.py."""
    from lumillyrender_amd import host
    desc = load("brdf-row.toml", 96, 54)               # hdr scene with values above 1
    scene = dev.Scene(desc)
    film = scene.render(desc.render_params(spp=16, seed=1))
    # RGBE: integer-exact on both sides -> identical bytes, identical files after decode
    rgbe = scene.quantize("rgbe")
    host.write_hdr_rgbe(tmp_path / "dev.hdr", rgbe)
    host.save_hdr(tmp_path / "host.hdr", film)
    assert np.array_equal(host.load_hdr(tmp_path / "dev.hdr"), host.load_hdr(tmp_path / "host.hdr"))
    # RGB8: powf on the host vs the device's own pow series: equal except for rare bucket-edge pixels (off by one)
    q = scene.quantize("rgb8", gamma=2.2)
    ref = host.to_color(film, 2.2)
    diff = np.abs(q.astype(int) - ref.astype(int))
    assert diff.max() <= 1 and (diff != 0).mean() < 2e-3
    host.write_png_rgb8(tmp_path / "dev.png", q)
    from PIL import Image
    assert np.array_equal(np.array(Image.open(tmp_path / "dev.png").convert("RGB")), q)
    scene.close()


@pytest.mark.parametrize("first", [0, 1000])
def test_random_scenes_match_oracle(dev, oracle, first):
    """tools/fuzz_parity.py: random scenes of 3..60 spheres and quads (flat loop and 4-wide tree), all five BSDFs with
    random parameters, area lights or sky, the three cameras, pt and pt-direct; the default pipeline and the streaming
    one against the oracle.  400 seeds were run when this was written (worst relative error 1.1e-6); a dozen stay here."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    for seed in range(first, first + 6):
        worst, n_prims, integ, cam, mean, nan = fz.run(seed, 40, 30, 16)
        assert worst < TOL, (seed, worst, n_prims, integ, cam)
    if first == 1000:
        # hostile corners (non-integral and huge Phong exponents, roughness 0 and 3, ior 1 / 0 / 1e5, black and unit albedos,
        # zero-area and needle quads, pin-head and planet spheres): equal non-finite masks, equal finite values
        n_nan = 0
        for seed in range(0, 12):
            worst, n_prims, integ, cam, mean, nan = fz.run(seed, 40, 30, 8, 26, True)
            assert worst < TOL, (seed, worst)
            n_nan += nan > 0
        assert n_nan >= 1                                 # the corners really produce lost samples
    if first == 0:
        # seed 400649 of a 24 000-seed run in round 2: a Blinn-Phong sample whose pdf underflows to 0 makes the reference's
        # recursion return 0 * c / 0 = NaN for the pixel whatever the rest of the path does; the throughput form has to
        # poison the sample too (identical NaN masks are part of fz.run's check)
        worst, n_prims, integ, cam, mean, nan = fz.run(400649, 40, 30, 8, 600)
        assert worst < TOL and nan > 0, (worst, nan)


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


@pytest.mark.parametrize("name,W,H", FULL_SIZE)
def test_full_size_properties_of_configs_3_to_5(dev, name, W, H):
    """BASELINE configs 3-5 at their FULL film sizes (4 spp; the oracle is too slow there): the device's finished-sample
    counter equals W*H*spp, every path statistic is positive, the film is finite (a GGX sample below the horizon has a
    negative cosine and the reference, ggx.rs:87-113 / scene.rs:99, does not clamp it: slightly negative pixels are its
    output too), and the frame rendered as four interleaved tile shards (what four GPUs would do) equals the untiled frame
    bit for bit."""
    if name != "brdf-row.toml" and not _generated_assets():
        pytest.skip("generated assets missing (run __graft_entry__.build())")
    from lumillyrender_amd import host
    desc = load(name, W, H)
    scene = dev.Scene(desc)
    p = desc.render_params(spp=4, seed=21)
    a = scene.render(p)
    st = scene.stats()
    assert st.samples == W * H * 4 and st.segments >= st.samples
    assert np.isfinite(a).all() and a.max() > 0 and a.min() > -1e-2
    out = np.zeros_like(a)
    shard_samples = 0
    for rank in range(4):
        tiles, n = host.tiles(W, H, 64, rank, 4)
        scene.render(p, tiles, n, out=out)
        shard_samples += scene.stats().samples
    assert shard_samples == W * H * 4
    assert np.array_equal(a, out)
    scene.close()


def test_golden_film_crops(dev):
    """tests/golden/*.npy (written by tests/golden/make_golden.py with the oracle, pinned on CPU by tests/test_golden_fixtures.py):
    every scene class of SURVEY 8c -- flat Lambert pt / pt-direct, the BRDF row with GGX, Phong and Blinn-Phong lobes, the
    100k-triangle mesh, thin lens + IBL -- through lr_render, against the committed crop, no oracle in the process."""
    from tests import golden_cases as gc
    ran = 0
    for case in gc.FILM_CASES:
        name, edit, w, h, spp, integ, seed, gen = case
        if gen and not gc.have_generated_assets():
            continue
        desc = gc.load_scene(name, edit, w, h)
        scene = dev.Scene(desc)
        img = scene.render(desc.render_params(spp=spp, seed=seed, integrator=integ))
        ref = np.load(os.path.join(gc.GOLDEN, gc.film_name(case)))
        assert np.array_equal(np.isnan(img), np.isnan(ref)), case           # powf(negative, non-integer) NaNs of the Phong lobes: same mask
        # HDR film (IBL texels of ~10^3): a pixel brighter than 1 gets 1e-4 of ITS OWN value -- per pixel, not of the film's maximum
        bar = TOL * np.maximum(1.0, np.abs(np.nan_to_num(ref))) if name == "ibl-lens.toml" else TOL
        assert np.all(np.abs(np.nan_to_num(img) - np.nan_to_num(ref)) < bar), case
        scene.close(); ran += 1
    assert ran >= 6


@pytest.mark.parametrize("case", STATED)
@pytest.mark.parametrize("film", [(4, 4), (8, 6)], ids=["4x4", "8x6"])
def test_stated_spp_parity_with_the_oracle(dev, oracle, case, film):
    """configs[1..4] at their stated 1024 / 4096 / 2048 / 8192 spp, default pipeline, against the oracle's flat fold of the same
    samples (main.rs:92-121) through the reference-literal tree (bvh.rs:131-141): the 1e-4 bar, identical sample / segment /
    shadow-ray / sky-fetch counters."""
    name, edit, _, _, spp, integ, seed, gen = case
    if gen and not gc.have_generated_assets():
        pytest.skip("generated assets missing")
    w, h = film
    desc = gc.load_scene(name, edit, w, h)
    p = desc.render_params(spp=spp, seed=seed + 1, integrator=integ)
    scene = dev.Scene(desc)
    img = scene.render(p)
    st = scene.stats()
    ref, so = oracle.render(desc, p, mode=oracle.BVH, pad=0.0, with_stats=True)
    assert np.isfinite(ref).all() and np.isfinite(img).all()
    assert (st.samples, st.segments, st.shadow_rays, st.sky_fetches) == (so.samples, so.segments, so.shadow_rays, so.sky_fetches)
    assert st.samples == w * h * spp
    assert _within_bar(img, ref).all(), float(np.max(np.abs(img - ref) / np.maximum(1.0, np.abs(ref))))
    scene.close()


def test_stated_spp_golden_crops(dev):
    """tests/golden/*_8x6_<stated spp>spp_*.npy (oracle output, pinned on CPU by tests/test_golden_fixtures.py) through lr_render
    with no oracle in the process."""
    ran = 0
    for case in gc.STATED_SPP_CASES:
        name, edit, w, h, spp, integ, seed, gen = case
        if gen and not gc.have_generated_assets():
            continue
        desc = gc.load_scene(name, edit, w, h)
        scene = dev.Scene(desc)
        img = scene.render(desc.render_params(spp=spp, seed=seed, integrator=integ))
        ref = np.load(os.path.join(gc.GOLDEN, gc.film_name(case)))
        assert _within_bar(img, ref).all(), (case, float(np.max(np.abs(img - ref))))
        assert scene.stats().samples == w * h * spp
        scene.close(); ran += 1
    assert ran >= 2


def test_hdr_golden_crop_per_pixel_bar(dev):
    """The thin-lens + IBL crop (film max 64, median 0.5) under the PER-PIXEL bar |img - ref| < 1e-4 max(1, |ref|): round 4 allowed
    every pixel 1e-4 of the film's maximum (6.4e-3)."""
    if not gc.have_generated_assets():
        pytest.skip("generated assets missing")
    case = [c for c in gc.FILM_CASES if c[0] == "ibl-lens.toml"][0]
    name, edit, w, h, spp, integ, seed, _ = case
    desc = gc.load_scene(name, edit, w, h)
    scene = dev.Scene(desc)
    img = scene.render(desc.render_params(spp=spp, seed=seed, integrator=integ))
    ref = np.load(os.path.join(gc.GOLDEN, gc.film_name(case)))
    assert float(ref.max()) > 10.0
    assert _within_bar(img, ref).all(), float(np.max(np.abs(img - ref) / np.maximum(1.0, np.abs(ref))))
    scene.close()


@pytest.mark.parametrize("spp", [1, 7, 15, 16, 17, 31, 32, 33, 63, 64, 100, 129, 257, 1000, 1025, 1040, 2047])
def test_chunk_schedule_edges(dev, oracle, spp):
    """The chunk schedule is a function of spp only (body of 8- / 16- / 32-sample chunks + a taper of halving lengths down to
    single samples, a shorter last body chunk when spp does not divide): every sample exactly once -- counters equal the oracle's,
    film within the bar -- for spp around every edge of the rule."""
    desc = gc.load_scene("cbox-spheres.toml", None, 6, 4)
    p = desc.render_params(spp=spp, seed=11, integrator=1)
    scene = dev.Scene(desc)
    img = scene.render(p)
    st = scene.stats()
    ref, so = oracle.render(desc, p, with_stats=True)
    assert st.samples == 6 * 4 * spp
    assert (st.samples, st.segments, st.shadow_rays) == (so.samples, so.segments, so.shadow_rays)
    assert _within_bar(img, ref).all()
    scene.close()


@pytest.mark.parametrize("scene_name,integ,flag_names", [("cbox-spheres.toml", 1, ("default", "resident", "streaming")),
                                                         ("brdf-row.toml", 1, ("default", "fused")),
                                                         ("mesh-box.toml", 0, ("default", "streaming")),
                                                         ("ibl-lens.toml", 1, ("default",))], ids=["cbox", "brdf", "mesh", "ibl"])
def test_pixel_bands_give_the_same_film(dev, knobs, scene_name, integ, flag_names):
    """A call whose chunk sums exceed 3 GiB is rendered in bands of consecutive pixel ranks, one launch each
    (Img::new is W x H whatever the spp, img.rs:13), and inside a launch the work items are dealt in sub-bands of 2^17 pixel ranks
    (the rays in flight stay within a strip of the film).  LR_BAND_PIX / LR_SUB_SHIFT force small bands and sub-bands on a small
    film: same film bits, same counters as the one-band render, in every pipeline -- including ragged last bands, a last sub-band
    that takes the remainder, and a tile list of several tiles."""
    monkeypatch = knobs                                  # (the knob build of the library: the product one reads no LR_* switch, csrc/lr_knobs.h)
    from lumillyrender_amd import abi, host
    if scene_name in ("mesh-box.toml", "ibl-lens.toml") and not gc.have_generated_assets():
        pytest.skip("generated assets missing")
    W, H, spp = 96, 72, 48
    desc = gc.load_scene(scene_name, None, W, H)
    flags = {"default": 0, "resident": abi.LR_FLAG_RESIDENT, "streaming": abi.LR_FLAG_STREAMING, "fused": abi.LR_FLAG_FUSED}
    tiles, n = host.tiles(W, H, 16, 1, 3)                                    # rank 1 of 3: a list of scattered 16-px tiles
    for fname in flag_names:
        p = desc.render_params(spp=spp, seed=9, integrator=integ, flags=flags[fname])
        monkeypatch.delenv("LR_BAND_PIX", raising=False)
        monkeypatch.setenv("LR_SUB_SHIFT", "0")
        one = dev.Scene(desc)
        ref = one.render(p); sr = one.stats()
        film_ref = np.full((H, W, 3), -1.0, dtype=np.float32); one.render(p, tiles, n, out=film_ref)
        one.close()
        for band, sub in ((None, 8), (None, 11), (1024, 7), (3072, 0), (5000, 9)):
            if band is None:
                monkeypatch.delenv("LR_BAND_PIX", raising=False)
            else:
                monkeypatch.setenv("LR_BAND_PIX", str(band))
            monkeypatch.setenv("LR_SUB_SHIFT", str(sub))
            sc = dev.Scene(desc)
            img = sc.render(p); st = sc.stats()
            assert np.array_equal(_bits(img), _bits(ref)), (fname, band, sub)
            assert (st.samples, st.segments, st.shadow_rays, st.sky_fetches) == (sr.samples, sr.segments, sr.shadow_rays, sr.sky_fetches), (fname, band, sub)
            film = np.full((H, W, 3), -1.0, dtype=np.float32); sc.render(p, tiles, n, out=film)
            assert np.array_equal(_bits(film), _bits(film_ref)), (fname, band, sub)
            img2 = sc.render(p)                                              # a second frame through the same scene (buffers reused)
            assert np.array_equal(_bits(img2), _bits(ref)), (fname, band, sub)
            sc.close()
    monkeypatch.delenv("LR_BAND_PIX", raising=False)
    monkeypatch.delenv("LR_SUB_SHIFT", raising=False)


def test_config5_at_its_stated_size_stays_under_three_gigabytes(dev):
    """the chunk sums of config 5 (2048 x 2048 at 8192 spp) were 17.2 GB in one buffer.  In pixel bands (at most
    2 GiB of sums each, one after the other) the whole call -- scene, film, chunk sums -- adds less than 3 GB of device memory.  One frame at the stated size
    (34 G samples, ~8 s): every sample rendered, film finite."""
    import ctypes as C
    if not gc.have_generated_assets():
        pytest.skip("generated assets missing")
    hip = C.CDLL("libamdhip64.so")
    free0, free1, total = C.c_size_t(), C.c_size_t(), C.c_size_t()
    desc = gc.load_scene("ibl-lens.toml", None, 2048, 2048)
    assert hip.hipMemGetInfo(C.byref(free0), C.byref(total)) == 0
    scene = dev.Scene(desc)
    img = scene.render(desc.render_params(spp=8192, seed=1))
    assert hip.hipMemGetInfo(C.byref(free1), C.byref(total)) == 0
    st = scene.stats()
    assert st.samples == 2048 * 2048 * 8192 and np.isfinite(img).all() and float(img.max()) > 1.0
    used = free0.value - free1.value
    assert used < 3 * (1 << 30), used / 2**30
    scene.close()


def test_more_than_two_to_the_32_work_items_in_one_call(dev):
    """Round 4 refused a call with more than 2^32 - 2^24 work items (pixels x chunks): a 4096^2 film at 8192 spp.  The limit is
    per pixel band now, and bands are cut by the chunk-sum budget, so the call goes through: 4096 x 3200 pixels x 336 chunks =
    4.4 * 10^9 work items, 1.07 * 10^11 samples of the Cornell scene (~18 s), every sample rendered; a 16 x 16 tile of it rendered
    on its own gives the same bits (the film does not depend on tiling or banding)."""
    from lumillyrender_amd import abi
    W, H, spp = 4096, 3200, 8192
    desc = gc.load_scene("cbox-spheres.toml", None, W, H)
    scene = dev.Scene(desc)
    p = desc.render_params(spp=spp, seed=4, integrator=1)
    img = scene.render(p)
    st = scene.stats()
    assert st.samples == W * H * spp
    assert np.isfinite(img).all() and float(img.max()) > 0.1
    tile = (abi.LrTile * 1)(); tile[0].x0, tile[0].y0, tile[0].w, tile[0].h = 2048, 1600, 16, 16
    small = np.zeros((H, W, 3), dtype=np.float32)
    scene.render(p, tile, 1, out=small)
    assert np.array_equal(_bits(small[1600:1616, 2048:2064]), _bits(img[1600:1616, 2048:2064]))
    scene.close()


@pytest.mark.parametrize("key", STATED_SIZE)
def test_stated_size_tiles_against_the_live_oracle(dev, oracle, key):
    """configs[0..4] (+ the Phong / Blinn-Phong variants of configs[2]) at their STATED film size and spp: lr_render on 64 scattered
    16-px tiles of the full-size film (+ whole rows through the box's edges for configs[1]) against the oracle's reference-literal
    mode (SAH tree, collect-all-candidates walk, first minimum: bvh.rs:20-25,38-45,131-141) on exactly those tiles -- per-pixel bar on every pixel, equal sample / segment / shadow-ray / sky-fetch counters."""
    desc, p, tl, img, st = _render_tiles(dev, key)
    mask = gc.tile_mask(p_w(desc), p_h(desc), tl)
    ref, so = oracle.render(desc, p, gc.tile_array(tl), len(tl), mode=oracle.BVH, pad=0.0, with_stats=True, fast=True)   # the reference line by line
    assert st.samples == int(mask.sum()) * p.spp == so.samples
    assert (st.segments, st.shadow_rays, st.sky_fetches) == (so.segments, so.shadow_rays, so.sky_fetches)
    ok = _within_bar(img[mask], ref[mask]) | (np.isnan(img[mask]) & np.isnan(ref[mask]))
    assert ok.all(), (int((~ok).sum()), float(np.nanmax(np.abs(img[mask] - ref[mask]))))
    assert np.all(img[~mask] == 0.0)                                   # only tile pixels are written


def test_a_denormal_pdf_keeps_its_sample_finite(dev, oracle):
    """phong.rs:47-68 at alpha = 20: a sampled direction almost across the lobe has pdf = 22 / 2 pi * c^20 in the DENORMAL range (not 0: that case is
    the reference's own NaN) and a BRDF value just as small; scene.rs:101 divides one by the other and gets a finite weight.  A 1-ulp hardware
    reciprocal reads a denormal as 0 -> inf -> NaN pixel.  The six pixels of the stated Phong row (960 x 540 x 4096 spp, seed 0) where the whole-frame
    check found it, each rendered up to and including the sample in question (RNG keys are (pixel, sample): a prefix of the stated samples)."""
    name, edit, w, h, spp, integ, seed, gen, rows = gc.STATED_SIZE_CASES["c3p"]
    desc = gc.load_scene(name, edit, w, h)
    scene = dev.Scene(desc)
    fixture = np.load(os.path.join(gc.GOLDEN, gc.DENORMAL_PDF_FIXTURE))          # the oracle's values, pinned on CPU by tests/test_golden_fixtures.py
    for i, (x, y, k) in enumerate(gc.DENORMAL_PDF_PIXELS):
        t = gc.one_pixel_tile(x, y)
        p = desc.render_params(spp=k + 1, seed=seed, integrator=integ)
        got = scene.render(p, t, 1)[y, x]
        ref = oracle.render(desc, p, t, 1, mode=oracle.BVH, pad=0.0)[y, x]
        assert np.isfinite(ref).all() and np.isfinite(got).all(), (x, y, k, got, ref)
        assert _within_bar(got, ref).all() and _within_bar(got, fixture[i]).all(), (x, y, k, got, ref, fixture[i])
    scene.close()


@pytest.mark.parametrize("key", STATED_SIZE)
def test_whole_stated_frame_against_the_live_oracle(dev, oracle, key):
    """EVERY pixel of a stated frame (main.rs:80-122 runs per pixel of the W x H film) against the oracle's reference-literal mode, at the stated
    spp: per-pixel bar, equal counters.  Minutes of host time per config (configs[1]: 1.07e9 samples, ~90 s on 16 cores; configs[4]: 3.4e10, ~50
    min), so opt-in: LUMILLY_WHOLE_FRAMES=c2,c3 selects the configs; LUMILLY_RECORD=<dir> keeps a one-line record of what was compared
    (profiles/r06_whole_frame_<config>.json are such records)."""
    import json, time
    if key not in os.environ.get("LUMILLY_WHOLE_FRAMES", "").split(","):
        pytest.skip("opt-in: LUMILLY_WHOLE_FRAMES=" + key)
    name, edit, w, h, spp, integ, seed, gen, rows = gc.STATED_SIZE_CASES[key]
    if gen and not gc.have_generated_assets():
        pytest.skip("generated assets missing")
    desc = gc.load_scene(name, edit, w, h)
    p = desc.render_params(spp=spp, seed=seed, integrator=integ)
    # LUMILLY_WHOLE_BAND="k/n": rows [k h / n, (k + 1) h / n) only -- a frame that takes an hour of oracle time in n calls
    band_k, band_n = (int(v) for v in os.environ.get("LUMILLY_WHOLE_BAND", "0/1").split("/"))
    y0, y1 = band_k * h // band_n, (band_k + 1) * h // band_n
    from lumillyrender_amd import abi
    t = (abi.LrTile * 1)(); t[0].x0, t[0].y0, t[0].w, t[0].h = 0, y0, w, y1 - y0
    scene = dev.Scene(desc)
    img = scene.render(p, t, 1)[y0:y1]
    st = scene.stats()
    dev_ms = st.render_ms
    scene.close()
    t0 = time.perf_counter()
    ref, so = oracle.render(desc, p, t, 1, threads=usable_cores(), mode=oracle.BVH, pad=0.0, with_stats=True, fast=True)
    ref = ref[y0:y1]
    cpu_s = time.perf_counter() - t0
    both_nan = np.isnan(img) & np.isnan(ref)
    ok = _within_bar(img, ref) | both_nan
    diff = np.where(both_nan, 0.0, np.abs(img - ref))
    rec = {"config": key, "scene": name, "edit": edit, "width": w, "height": h, "spp": spp, "integrator": integ, "seed": seed,
           "rows": [y0, y1], "pixels": w * (y1 - y0), "samples": int(st.samples), "oracle_mode": "BVH (reference-literal: SAH tree, every overlapped leaf whose own box passes, first minimum), pad 0",
           "pixels_over_the_bar": int((~ok).any(axis=2).sum()), "pixels_differing_at_all": int((_bits(img) != _bits(ref)).any(axis=2).sum()),
           "pixels_nan_in_both": int(both_nan.any(axis=2).sum()), "max_abs_diff": float(np.nanmax(diff)),
           "max_diff_over_bar_unit": float(np.nanmax(diff / (TOL * np.maximum(1.0, np.abs(np.where(both_nan, 1.0, ref)))))),
           "counters_device": [int(x) for x in _counters(st)], "counters_oracle": [int(so.samples), int(so.segments), int(so.shadow_rays), int(so.sky_fetches)],
           "device_render_ms": round(dev_ms, 2), "oracle_seconds": round(cpu_s, 1), "oracle_threads": usable_cores(), "build": dev.build_info()}
    out = os.environ.get("LUMILLY_RECORD")
    if out:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, f"whole_frame_{key}.json" if band_n == 1 else f"whole_frame_{key}_band{band_k}of{band_n}.json"), "w") as f:
            f.write(json.dumps(rec) + "\n")
    assert rec["counters_device"] == rec["counters_oracle"], rec
    assert ok.all(), rec


@pytest.mark.parametrize("key", STATED_SIZE)
def test_stated_size_tiles_against_the_fixtures(dev, key):
    """The same renders against the committed fixtures (tests/golden/stated_<config>.npy: the tiles' pixels in tile order, written by
    tests/golden/make_golden.py with the oracle) -- no oracle in the process."""
    path = os.path.join(gc.GOLDEN, gc.stated_name(key))
    if not os.path.exists(path):
        pytest.skip("fixture missing (tests/golden/make_golden.py)")
    desc, p, tl, img, st = _render_tiles(dev, key)
    mask = gc.tile_mask(p_w(desc), p_h(desc), tl)
    ref = np.load(path)
    got = gc.pack_tiles(img, tl)
    assert got.shape == ref.shape
    ok = _within_bar(got, ref) | (np.isnan(got) & np.isnan(ref))
    assert ok.all(), (int((~ok).sum()), float(np.nanmax(np.abs(got - ref))))
    assert st.samples == int(mask.sum()) * p.spp
