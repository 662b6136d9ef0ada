"""GPU parity -- pipelines.

The three pipelines (fused: a lane carries its path; resident: state in LDS; streaming: one launch per stage) and their variants give the same bits: same device
functions, same RNG keys, same chunk order, same order of additions into a sample's radiance.

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


@pytest.mark.parametrize("name,integ", [("cbox-spheres.toml", 1), ("cbox-spheres.toml", 0), ("brdf-row.toml", 1), ("two-spheres.toml", None)])
def test_resident_and_streaming_pipelines_are_bit_identical(dev, oracle, name, integ):
    """The single-launch resident pipeline (path state in LDS) and the multi-kernel streaming pipeline
    (path state in HBM) run the same device functions with the same RNG keys and chunk order: same film,
    same path statistics, and both within tolerance of the oracle."""
    from lumillyrender_amd import abi
    desc = load(name, 56, 40)
    scene = dev.Scene(desc)
    films, stats = [], []
    for flags in (abi.LR_FLAG_RESIDENT, abi.LR_FLAG_STREAMING, abi.LR_FLAG_STREAMING | abi.LR_FLAG_COUNT):
        p = desc.render_params(spp=24, seed=13, integrator=integ, flags=flags)
        films.append(scene.render(p))
        st = scene.stats()
        stats.append((st.samples, st.segments, st.shadow_rays, st.kernel_launches[abi.LR_K_RESIDENT]))
    assert stats[0][3] == 1 and stats[1][3] == 0            # the flags really selected the two pipelines
    assert stats[0][:3] == stats[1][:3] == stats[2][:3]
    assert np.array_equal(films[0], films[1]) and np.array_equal(films[1], films[2])
    ref = oracle.render(desc, desc.render_params(spp=24, seed=13, integrator=integ))
    assert linf(films[0], ref) < TOL
    scene.close()


def test_streaming_long_passes_lose_no_work(dev):
    """With millions of path slots a k_trace workgroup pass spans many segments and k_shade shades whole ranges from
    one work-item pool.  Every work item must still be rendered exactly once: the finished-sample counter equals
    W*H*spp and the film is bit-identical to a 4096-slot render (an earlier version stranded items in pools of
    list slices that ran empty near the end of the render: 10 % of the samples missing, nothing else wrong)."""
    from lumillyrender_amd import abi
    W, H, spp = 192, 128, 512                       # 24576 pixels x 64 chunks = 1.6 M work items
    desc = load("mesh-box.toml", W, H)
    scene = dev.Scene(desc)
    films = []
    for slots in (4096, 1 << 20, 0):
        img = scene.render(desc.render_params(spp=spp, seed=9, flags=abi.LR_FLAG_STREAMING, path_slots=slots))
        assert scene.stats().samples == W * H * spp, (slots, scene.stats().samples)
        films.append(img)
    assert np.array_equal(films[0], films[1]) and np.array_equal(films[0], films[2])
    desc2 = load("ibl-lens.toml", W, H)             # shadow lists ride on the same ranges
    scene2 = dev.Scene(desc2)
    a = scene2.render(desc2.render_params(spp=256, seed=2, flags=abi.LR_FLAG_STREAMING, path_slots=4096))
    n_a = scene2.stats().samples
    b = scene2.render(desc2.render_params(spp=256, seed=2, flags=abi.LR_FLAG_STREAMING, path_slots=0))
    assert n_a == scene2.stats().samples == W * H * 256
    assert np.array_equal(a, b)
    scene.close(); scene2.close()


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


@pytest.mark.parametrize("name,integ", FLAT_CASES)
def test_fused_pipeline_is_bit_identical_on_flat_scenes(dev, oracle, name, integ):
    """k_path_flat pairs the visibility test of a vertex's direct-light connection with the closest-hit test of the
    continuation ray (one pass over the primitive rows, origin-only terms shared) and keeps the path in registers.  Every
    operation that decides or rounds anything keeps its operands and order, so film and counters equal the resident
    and the streaming pipeline's bit for bit, and all of them sit within tolerance of the oracle."""
    from lumillyrender_amd import abi
    desc = load(name, 72, 40)
    scene = dev.Scene(desc)
    films, stats = {}, {}
    for label, flags in (("fused", abi.LR_FLAG_FUSED), ("resident", abi.LR_FLAG_RESIDENT), ("streaming", abi.LR_FLAG_STREAMING)):
        films[label] = scene.render(desc.render_params(spp=24, seed=13, integrator=integ, flags=flags))
        st = scene.stats()
        stats[label] = _counters(st)
        assert st.pipeline == {"fused": 2, "resident": 1, "streaming": 0}[label], (label, st.pipeline)
    assert stats["fused"] == stats["resident"] == stats["streaming"]
    assert stats["fused"][0] == 72 * 40 * 24
    assert np.array_equal(films["fused"], films["resident"]) and np.array_equal(films["fused"], films["streaming"])
    ref = oracle.render(desc, desc.render_params(spp=24, seed=13, integrator=integ))
    assert float(np.nanmax(np.abs(films["fused"] - ref))) < TOL
    scene.close()


def test_fused_pipeline_edge_sizes(dev):
    """Fewer work items than lanes, one pixel, a ragged tile list, spp that does not divide into chunks, and a frame with
    more work items than one wave's pool batch: every sample is rendered exactly once (device counter) and the film equals
    the resident pipeline's."""
    from lumillyrender_amd import abi, host
    for (w, h, spp) in ((1, 1, 1), (3, 2, 7), (16, 16, 3), (64, 48, 100), (200, 120, 37)):
        desc = load("cbox-spheres.toml", w, h)
        scene = dev.Scene(desc)
        a = scene.render(desc.render_params(spp=spp, seed=3, flags=abi.LR_FLAG_FUSED))
        st = scene.stats()
        assert st.pipeline == 2 and st.samples == w * h * spp, (w, h, spp, st.samples)
        b = scene.render(desc.render_params(spp=spp, seed=3, flags=abi.LR_FLAG_RESIDENT))
        assert np.array_equal(a, b), (w, h, spp)
        scene.close()
    desc = load("cbox-spheres.toml", 96, 64)
    scene = dev.Scene(desc)
    p = desc.render_params(spp=20, seed=8, flags=abi.LR_FLAG_FUSED)
    whole = scene.render(p)
    out = np.zeros_like(whole)
    for rank in range(3):
        tiles, n = host.tiles(96, 64, 32, rank, 3)
        scene.render(p, tiles, n, out=out)
    assert np.array_equal(whole, out)
    scene.close()


def test_flat_fused_kernel_spare_queues(dev, oracle):
    """k_path_flat keeps two spare camera samples per lane (one beside a thin lens' aperture points; the host picks the
    instantiation).  Both queue depths, frames with fewer work items than lanes and chunk lengths that do not divide spp: every
    sample once, film equal to the resident pipeline's bit for bit and within tolerance of the oracle."""
    from lumillyrender_amd import abi
    lens = lambda t: t.replace('type = "ideal-pinhole"\nfov = 39.3077', 'type = "thin-lens"\nfov = 39.3077\nfocus-distance = 800\nf-number = 2.8')
    for edit in (None, lens):
        for (w, h, spp) in ((2, 1, 5), (33, 17, 19), (96, 64, 24)):
            desc = load("cbox-spheres.toml", w, h, edit)
            assert edit is None or "thin-lens" in lens(open(scene_path("cbox-spheres.toml")).read())
            scene = dev.Scene(desc)
            a = scene.render(desc.render_params(spp=spp, seed=21, flags=abi.LR_FLAG_FUSED))
            st = scene.stats()
            assert st.pipeline == 2 and st.samples == w * h * spp, (w, h, spp, st.samples)
            b = scene.render(desc.render_params(spp=spp, seed=21, flags=abi.LR_FLAG_RESIDENT))
            assert np.array_equal(a, b), (edit is not None, w, h, spp)
            if (w, h) == (33, 17):
                ref = oracle.render(desc, desc.render_params(spp=spp, seed=21))
                assert float(np.nanmax(np.abs(a - ref))) < TOL
            scene.close()


@pytest.mark.parametrize("name,spp", [("mesh-box.toml", 24), ("ibl-lens.toml", 16)])
def test_fused_pipeline_is_bit_identical_on_tree_scenes(dev, oracle, name, spp):
    """k_path_tree: the lane walks its own connection and continuation rays through the 4-wide tree and is shaded at the
    wave's retire points.  Same device functions, RNG keys and order of additions as the streaming pipeline: same film, same
    counters; and within the stated tolerance of the oracle (IBL: relative to the film's range)."""
    if not _generated_assets():
        pytest.skip("generated assets missing (run __graft_entry__.build())")
    from lumillyrender_amd import abi
    desc = load(name, 160, 120)
    scene = dev.Scene(desc)
    a = scene.render(desc.render_params(spp=spp, seed=41, flags=abi.LR_FLAG_FUSED))
    sa = scene.stats()
    b = scene.render(desc.render_params(spp=spp, seed=41, flags=abi.LR_FLAG_STREAMING))
    sb = scene.stats()
    assert sa.pipeline == 2 and sb.pipeline == 0
    assert _counters(sa) == _counters(sb) and sa.samples == 160 * 120 * spp
    assert np.array_equal(a, b)
    small = load(name, 48, 36)
    s2 = dev.Scene(small)
    img = s2.render(small.render_params(spp=8, seed=5, flags=abi.LR_FLAG_FUSED))
    ref = oracle.render(small, small.render_params(spp=8, seed=5))
    fin = np.isfinite(ref)
    assert np.array_equal(np.isnan(img), np.isnan(ref)) and np.all(np.abs(img[fin] - ref[fin]) < TOL * np.maximum(1.0, np.abs(ref[fin])))
    scene.close(); s2.close()


@pytest.mark.parametrize("name", ["mesh-box.toml", "ibl-lens.toml"])
def test_spare_camera_samples_edge_sizes_on_tree_scenes(dev, name):
    """The finish stage of the fused kernels hands every lane its next camera sample from LDS (path_spare_batch /
    path_consume, thin lens: with the aperture point).  One pixel, fewer work items than lanes of a wave, chunk lengths that do
    not divide spp, chunk ends inside a batch, and a three-rank tile split: every sample exactly once (device counter), film
    equal to the streaming pipeline's bit for bit."""
    if not _generated_assets():
        pytest.skip("generated assets missing (run __graft_entry__.build())")
    from lumillyrender_amd import abi, host
    for (w, h, spp) in ((1, 1, 1), (1, 1, 19), (5, 3, 17), (40, 30, 33), (64, 2, 9)):
        desc = load(name, w, h)
        scene = dev.Scene(desc)
        a = scene.render(desc.render_params(spp=spp, seed=11, flags=abi.LR_FLAG_FUSED))
        st = scene.stats()
        assert st.pipeline == 2 and st.samples == w * h * spp, (name, w, h, spp, st.samples)
        b = scene.render(desc.render_params(spp=spp, seed=11, flags=abi.LR_FLAG_STREAMING))
        assert np.array_equal(a, b, equal_nan=True), (name, w, h, spp)
        scene.close()
    desc = load(name, 72, 40)
    scene = dev.Scene(desc)
    p = desc.render_params(spp=12, seed=2, flags=abi.LR_FLAG_FUSED)
    whole = scene.render(p)
    out = np.zeros_like(whole)
    for rank in range(3):
        tiles, n = host.tiles(72, 40, 8, rank, 3)
        scene.render(p, tiles, n, out=out)
    assert np.array_equal(whole, out, equal_nan=True)
    scene.close()


def test_fused_pipeline_on_random_tree_scenes(dev, oracle):
    """Random scenes with more than 32 primitives (spheres and transformed quads, all five BSDFs, area lights or sky, the
    three cameras, both integrators) through k_path_tree<31>: film equal to the streaming pipeline's bit for bit, equal NaN
    masks with the oracle and the tolerance on every finite value."""
    import importlib.util
    from lumillyrender_amd import abi, host
    spec = importlib.util.spec_from_file_location("fz", os.path.join(ROOT, "tools", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
    done = 0
    for seed in range(3000, 3040):
        text, integ, cam = fz.scene_text(seed, 40, 30, 90)
        desc = host.Description(text=text)
        if desc.desc.n_prims <= 32:
            continue
        scene = dev.Scene(desc)
        a = scene.render(desc.render_params(spp=8, seed=seed, flags=abi.LR_FLAG_FUSED))
        sa = scene.stats()
        b = scene.render(desc.render_params(spp=8, seed=seed, flags=abi.LR_FLAG_STREAMING))
        sb = scene.stats()
        assert sa.pipeline == 2 and _counters(sa) == _counters(sb), seed
        assert np.array_equal(a, b, equal_nan=True), seed
        ref = oracle.render(desc, desc.render_params(spp=8, seed=seed))
        assert np.array_equal(np.isnan(a), np.isnan(ref)), seed
        fin = np.isfinite(ref) & np.isfinite(a)
        assert float(np.max(np.abs(a[fin] - ref[fin]) / np.maximum(1.0, np.abs(ref[fin])), initial=0.0)) < TOL, seed
        scene.close()
        done += 1
    assert done >= 10


@pytest.mark.parametrize("name,W,H", FULL_SIZE[1:])
def test_full_size_film_is_pipeline_independent(dev, name, W, H):
    """Configs 4 and 5 at full film size (2 spp): the fused kernel (one launch, ~1800 workgroups that draw work items from
    per-wave pools in whatever order they get to them) and the streaming pipeline (millions of slots, three launches per
    iteration) produce the same film bit for bit and the same counters -- the film depends on neither scheduling nor pipeline."""
    if not _generated_assets():
        pytest.skip("generated assets missing (run __graft_entry__.build())")
    from lumillyrender_amd import abi
    desc = load(name, W, H)
    scene = dev.Scene(desc)
    a = scene.render(desc.render_params(spp=2, seed=33, flags=abi.LR_FLAG_FUSED))
    sa = scene.stats()
    b = scene.render(desc.render_params(spp=2, seed=33, flags=abi.LR_FLAG_STREAMING))
    sb = scene.stats()
    assert sa.pipeline == 2 and sb.pipeline == 0 and _counters(sa) == _counters(sb) and sa.samples == W * H * 2
    assert np.array_equal(a, b)
    scene.close()


@pytest.mark.gpu
def test_fused_kernels_get_the_workgroups_per_cu_they_are_sized_for(dev, monkeypatch, capfd):
    """The fused kernels launch one workgroup per resident slot; a kernel whose LDS creeps over a 1280-B granule boundary loses a
    workgroup per CU (tools/micro/lds_granule.hip), and the render then waits for the workgroups that did not fit (-6 % when it
    happened).  lr_render prices the LDS itself and says what fits under LR_DEBUG: 6 for the flat and the pt-direct tree kernels
    (thin lens included), 7 for the pt tree kernel."""
    if not _generated_assets():
        pytest.skip("generated assets missing (run __graft_entry__.build())")
    import re
    monkeypatch.setenv("LR_DEBUG", "1")
    for name, want in (("cbox-spheres.toml", 6), ("mesh-box.toml", 7), ("ibl-lens.toml", 6)):
        desc = load(name, 64, 48)
        scene = dev.Scene(desc)
        capfd.readouterr()
        scene.render(desc.render_params(spp=2, seed=1))
        err = capfd.readouterr().err
        m = re.search(r"fused kernel: (\d+) workgroups per CU fit \((\d+) wanted\)", err)
        assert m, err[-400:]
        assert int(m.group(1)) >= int(m.group(2)) == want, (name, m.group(0))
        scene.close()


def test_path_slots_bound_and_resident_request_that_does_not_fit(dev):
    """LrRenderParams.path_slots is an upper bound for the fused pipeline (it used to be ignored), and LR_FLAG_RESIDENT on a scene
    whose traversal stack does not fit the LDS beside the state now runs the default (fused) pipeline instead of dropping to the
    streaming one.  Neither changes a film (RNG keys are scheduling-independent)."""
    if not _generated_assets():
        pytest.skip("generated assets missing")
    from lumillyrender_amd import abi
    desc = load("mesh-box.toml", 96, 64)
    scene = dev.Scene(desc)
    base = scene.render(desc.render_params(spp=8, seed=9)); st0 = scene.stats()
    assert st0.pipeline == 2 and st0.path_slots > 4096
    small = scene.render(desc.render_params(spp=8, seed=9, path_slots=3000)); st1 = scene.stats()
    assert st1.pipeline == 2 and st1.path_slots == 3072                    # rounded up to whole 256-lane workgroups (and 512-slot segments)
    assert np.array_equal(base.view(np.uint32), small.view(np.uint32))
    res = scene.render(desc.render_params(spp=8, seed=9, flags=abi.LR_FLAG_RESIDENT)); st2 = scene.stats()
    assert st2.pipeline in (1, 2)                                           # resident if it fits this scene's stack, else the default -- never streaming
    assert np.array_equal(base.view(np.uint32), res.view(np.uint32))
    scene.close()


@pytest.mark.parametrize("case", STATED)
def test_stated_spp_films_are_pipeline_independent(dev, case):
    """At the stated spp the default pipeline, the streaming pipeline and (flat scenes) the resident and the fused one give the
    same bits: same chunk schedule, same order of additions into a sample's radiance, same k_resolve."""
    from lumillyrender_amd import abi
    name, edit, _, _, spp, integ, seed, gen = case
    if gen and not gc.have_generated_assets():
        pytest.skip("generated assets missing")
    desc = gc.load_scene(name, edit, 4, 4)
    scene = dev.Scene(desc)
    films, stats = {}, {}
    flags = {"default": 0, "streaming": abi.LR_FLAG_STREAMING, "fused": abi.LR_FLAG_FUSED}
    if name in ("cbox-spheres.toml", "brdf-row.toml"):
        flags["resident"] = abi.LR_FLAG_RESIDENT
    for k, f in flags.items():
        films[k] = scene.render(desc.render_params(spp=spp, seed=seed, integrator=integ, flags=f))
        s = scene.stats()
        stats[k] = (s.samples, s.segments, s.shadow_rays, s.sky_fetches, s.pipeline)
    assert stats["streaming"][4] == 0 and stats["fused"][4] == 2
    for k in flags:
        assert np.array_equal(_bits(films[k]), _bits(films["default"])), k
        assert stats[k][:4] == stats["default"][:4], k
    scene.close()
