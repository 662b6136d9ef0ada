"""GPU parity, round 3: the fused pipeline (csrc/lr_path.h: one persistent launch, a lane carries its path) against the
other two pipelines and the oracle; full-size count / tiling properties of BASELINE configs 3-5; the RCCL branch of
bench.py."""
import json
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


def _generated_assets():
    return os.path.exists(os.path.join(ROOT, "assets", "models", "blob", "blob.obj"))


def _counters(st):
    return (st.samples, st.segments, st.shadow_rays, st.sky_fetches)


FLAT_CASES = [("cbox-spheres.toml", 1), ("cbox-spheres.toml", 0), ("brdf-row.toml", 1), ("brdf-row.toml", 0), ("two-spheres.toml", None)]


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


FULL_SIZE = [("brdf-row.toml", 960, 540), ("mesh-box.toml", 1920, 1370), ("ibl-lens.toml", 2048, 2048)]


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


def test_bench_rccl_barrier_branch_runs(tmp_path):
    """bench.py's N > 1 branch -- gloo default group, RCCL sub-group, all-reduce barrier around the timed region -- executed
    on this box's one GPU (BENCH_FORCE_DIST=1, world 1), so that the driver's N-GPU run is not its first execution; the JSON
    line says which barrier bracketed the timed region."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--width", "192", "--height", "128",
                        "--spp", "16", "--no-cpu-baseline", "--backend", "nccl", "--tile", "32"], capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["barrier"] == "RCCL all-reduce + device synchronize", (line["barrier"], r.stderr[-1500:])
    assert line["n_gpus"] == 1 and line["value"] > 0 and "other_configs" not in line


@pytest.mark.parametrize("builder", ["ploc", "lbvh"])
def test_device_builders_prune_only(dev, oracle, knobs, builder):
    """the device builds the tree itself when the description carries none -- PLOC (default: bottom-up merging of
    Morton-ordered clusters, subtrees collapsed into leaves by SAH cost; within 2 % of the host SAH tree's render rate, 3 ms for
    10^5 triangles) or the round-1 Morton LBVH (LR_DEVICE_BVH=lbvh).  A tree only prunes: closest hits on random and grazing rays
    equal the device's brute force over all primitives, the film equals the host-tree film bit for bit, for a mesh, a handful
    of spheres, and the 2- and 3-primitive corner cases."""
    monkeypatch = knobs                                  # (the knob build of the library: the product one reads no LR_* switch, csrc/lr_knobs.h)
    if not _generated_assets():
        pytest.skip("generated assets missing (run __graft_entry__.build())")
    monkeypatch.setenv("LR_DEVICE_BVH", builder)
    rng = np.random.default_rng(31)
    desc = load("mesh-box.toml", 96, 64)
    host_scene, dev_scene = dev.Scene(desc), dev.Scene(desc, device_bvh=True)
    assert dev_scene.stats().bvh_build_ms > 0 and host_scene.stats().bvh_build_ms == 0
    p = desc.render_params(spp=12, seed=4)
    assert np.array_equal(host_scene.render(p), dev_scene.render(p))
    n = 400_000
    o = (rng.random((n, 3)) * 400 + 80).astype(np.float32)          # inside the box the mesh stands in
    d = rng.standard_normal((n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    tp, tt = dev_scene.intersect(o, d)
    bp, bt = dev_scene.intersect(o, d, brute=True)
    assert np.array_equal(tp, bp) and np.array_equal(tt, bt)
    assert (bp >= 0).mean() > 0.5
    host_scene.close(); dev_scene.close()
    for name in ("two-spheres.toml", "cbox-spheres.toml", "brdf-row.toml"):
        dsc = load(name, 40, 30)
        a, b = dev.Scene(dsc), dev.Scene(dsc, device_bvh=True)
        q = dsc.render_params(spp=6, seed=9)                       # (flat scenes test every primitive; the tree is still built, collapsed and validated by lr_scene_create)
        assert np.array_equal(a.render(q), b.render(q)), name
        a.close(); b.close()


def test_ploc_falls_back_on_needle_meshes(dev, monkeypatch, capfd):
    """A mesh stretched 40:1 (tools/fuzz_traversal.py seed 515) makes PLOC's area-driven merging chain up to a height of ~100;
    the builder then discards that tree for the radix tree, whose height the key length bounds, instead of refusing the scene.
    Either way the tree only prunes: every ray gets brute force's primitive and distance bits."""
    if not _generated_assets():
        pytest.skip("generated assets missing (run __graft_entry__.build())")
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_traversal", os.path.join(ROOT, "tools", "fuzz_traversal.py"))
    ft = importlib.util.module_from_spec(spec); spec.loader.exec_module(ft)
    monkeypatch.setenv("LR_DEBUG", "1")
    bad, excused, hit, n_prims, *_ = ft.run(515, 60_000)
    err = capfd.readouterr().err
    assert "falling back to the radix tree" in err, err[-600:]
    assert bad == [0, 0] and excused == 0 and hit > 0.3


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
