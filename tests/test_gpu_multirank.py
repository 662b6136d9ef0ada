"""GPU parity -- multirank.

More than one rank on ONE GPU: bench.py exactly as the driver launches its N-GPU run (torch.distributed.run), the film assembled from the ranks' tiles equals the
single-rank film, the RCCL barrier branch, the balance of the diagonal tile deal on the stated films.

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


@pytest.mark.parametrize("config,w,h", [("c2", 256, 256), ("c4", 480, 344)])
def test_eight_rank_bench_assembles_the_single_rank_film(tmp_path, config, w, h):
    """bench.py exactly as the driver launches its 8-GPU scaling run (torch.distributed.run, one process per rank, tiles
    i % 8), all eight ranks on this box's one GPU: eight scene uploads, eight host BVH builds, eight mappers of one /dev/shm
    film.  The assembled film equals the 1-rank film bit for bit, the line carries every rank's upload / build time, and no
    shared-memory segment is left behind (main.rs:61-65,129-132: the reference's channel drain, across processes)."""
    import glob
    if config == "c4" and not _generated_assets():
        pytest.skip("generated assets missing (run __graft_entry__.build())")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    before = set(glob.glob("/dev/shm/lumilly_film_*"))
    common = ["--config", config, "--steps", "1", "--warmup", "0", "--width", str(w), "--height", str(h), "--spp", "16", "--no-cpu-baseline", "--tile", "32"]
    one = tmp_path / "one.npy"
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dump-film", str(one)] + common,
                        capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert r1.returncode == 0, r1.stderr[-2000:]
    eight = tmp_path / "eight.npy"
    r8 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                         "--master-port", "29561", os.path.join(ROOT, "bench.py"), "--gpus", "8", "--same-device", "--backend", "gloo",
                         "--dump-film", str(eight)] + common, capture_output=True, text=True, cwd=ROOT, env=env, timeout=1200)
    assert r8.returncode == 0, r8.stderr[-3000:]
    a, b = np.load(one), np.load(eight)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and a.max() > 0
    line = json.loads([l for l in r8.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 8 and line["scaling"] == "strong" and line["config"]["spp"] == 16
    assert len(line["rank_upload_ms"]) == 8 and all(x > 0 for x in line["rank_upload_ms"])
    assert len(line["rank_host_bvh_build_s"]) == 8
    assert line["rank_render_ms"]["max"] >= line["rank_render_ms"]["min"] > 0
    # every rank's share of the frame as per-rank lists (a SCALE run that falls short must say where)
    ranks = line["ranks"]
    for key in ("render_ms", "dominant_kernel", "dominant_kernel_ms", "render_call_ms", "readback_ms", "barrier_wait_ms", "host_bvh_build_s", "upload_ms", "pixels", "rays"):
        assert len(ranks[key]) == 8, key
    assert sum(ranks["pixels"]) == w * h and all(r > 0 for r in ranks["rays"]) and all(m > 0 for m in ranks["render_ms"])
    assert all(c >= m for c, m in zip(ranks["render_call_ms"], ranks["render_ms"]))        # the call contains the device work
    assert max(ranks["rays"]) <= 1.5 * (sum(ranks["rays"]) / 8)                              # the deal is even (32-px tiles of a small film)
    assert line["env_overrides"] == {k: v for k, v in env.items() if k.startswith("LR_")}
    assert set(glob.glob("/dev/shm/lumilly_film_*")) <= before


@pytest.mark.parametrize("cfg", [("cbox-spheres.toml", 1024, 1024, 1), ("brdf-row.toml", 960, 540, 1), ("mesh-box.toml", 1920, 1370, 0),
                                 ("ibl-lens.toml", 2048, 2048, 1)], ids=["c2", "c3", "c4", "c5"])
def test_tile_deal_is_balanced_on_the_stated_films(dev, cfg):
    """lr_host_tiles at world 8 on the four stated films (4 spp): the heaviest rank's segments + shadow rays are within 3 % of the
    mean (round 4's `id % world` on 64-px tiles: +12 % / +12 % / +1 % / +5 %), and the eight shares assemble the one-rank film
    bit for bit."""
    from lumillyrender_amd import host
    name, W, H, integ = cfg
    if name in ("mesh-box.toml", "ibl-lens.toml") and not gc.have_generated_assets():
        pytest.skip("generated assets missing")
    desc = gc.load_scene(name, None, W, H)
    scene = dev.Scene(desc)
    p = desc.render_params(spp=4, seed=2, integrator=integ)
    whole = scene.render(p)
    film = np.full((H, W, 3), -1.0, dtype=np.float32)
    work = []
    for r in range(8):
        tiles, n = host.tiles(W, H, 0, r, 8)
        scene.render(p, tiles, n, out=film)
        s = scene.stats()
        work.append(int(s.segments) + int(s.shadow_rays))
    assert np.array_equal(_bits(film), _bits(whole))
    work = np.array(work, dtype=np.float64)
    assert work.max() / work.mean() <= 1.03, work / work.mean()
    assert work.min() / work.mean() >= 0.97, work / work.mean()
    scene.close()
