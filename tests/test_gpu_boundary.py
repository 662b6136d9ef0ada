"""GPU parity -- boundary.

The C ABI as a boundary: bad input answered with codes, several scene handles at once, a film shared between handles, the stand-alone C++ driver (counterpart of
main.rs:43-145).

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


def test_device_api_rejects_bad_input(dev):
    """Error behaviour at the boundary: the reference panics (main.rs:48,124,166; description.rs:34-178),
    the C ABI returns LR_E* codes with a message and stays usable."""
    import ctypes as C
    from lumillyrender_amd import abi, host
    desc = load("cbox-spheres.toml", 16, 16)
    d = desc.desc
    h = C.c_void_p()
    lib = dev.lib()
    # wrong ABI version / no such device
    bad = abi.LrSceneDesc.from_buffer_copy(d)
    bad.abi_version = 99
    assert lib.lr_scene_create(0, C.byref(bad), C.byref(h)) == abi.LR_EINVAL and b"abi_version" in lib.lr_last_error()
    assert lib.lr_scene_create(4096, desc.desc_ptr, C.byref(h)) == abi.LR_EINVAL
    # a BVH that does not cover every primitive
    bad = abi.LrSceneDesc.from_buffer_copy(d)
    bad.n_prims = d.n_prims - 1
    assert lib.lr_scene_create(0, C.byref(bad), C.byref(h)) == abi.LR_EINVAL
    # material index out of range
    prims = (abi.LrPrimitive * d.n_prims)()
    C.memmove(prims, d.prims, C.sizeof(prims))
    prims[3].material = 1000
    bad = abi.LrSceneDesc.from_buffer_copy(d)
    bad.prims = C.cast(prims, C.POINTER(abi.LrPrimitive))
    assert lib.lr_scene_create(0, C.byref(bad), C.byref(h)) == abi.LR_EINVAL
    # a triangle whose edges leave the range in which the device's 1/det is proven to be the IEEE quotient
    tri = next(i for i in range(d.n_prims) if d.prims[i].type == abi.LR_PRIM_TRIANGLE)
    C.memmove(prims, d.prims, C.sizeof(prims))
    for k in (3, 6):
        prims[tri].v[k] = 3.0e30
    assert lib.lr_scene_create(0, C.byref(bad), C.byref(h)) == abi.LR_EUNSUPPORTED and b"2^120" in lib.lr_last_error()
    # render-time errors leave the scene usable
    scene = dev.Scene(desc)
    for kw in ({"spp": 0}, {"spp": -3}):
        with pytest.raises(host.LumillyError):
            scene.render(desc.render_params(**kw))
    p = desc.render_params(spp=2)
    p.integrator = 7
    with pytest.raises(host.LumillyError):
        scene.render(p)
    with pytest.raises(host.LumillyError):
        scene.quantize("rgb8", gamma=0.0)
    img = scene.render(desc.render_params(spp=2, seed=1))
    assert np.isfinite(img).all() and img.max() > 0
    scene.close()
    scene.close()                                        # double close is harmless


def test_two_scenes_render_concurrently(dev, oracle):
    """Different LrScene handles may be driven from different host threads (one stream each)."""
    import threading
    descs = [load("cbox-spheres.toml", 48, 48), load("brdf-row.toml", 64, 36)]
    scenes = [dev.Scene(d) for d in descs]
    out = [None, None]

    def work(i):
        for _ in range(3):
            out[i] = scenes[i].render(descs[i].render_params(spp=16, seed=4))
    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for i in range(2):
        assert linf(out[i], oracle.render(descs[i], descs[i].render_params(spp=16, seed=4))) < TOL
        scenes[i].close()


def test_shared_film_buffer_across_scene_handles(dev):
    """lr_render writes only the pixels of the tiles it is given, so several handles (one per GPU in
    production, three on this GPU here) can fill ONE film from concurrent host threads -- what bench.py's ranks do
    with the film in /dev/shm.  The result equals a single full-frame render bit for bit."""
    import threading
    from lumillyrender_amd import host
    W, H, world = 160, 96, 3
    desc = load("cbox-spheres.toml", W, H)
    params = desc.render_params(spp=8, seed=6)
    one = dev.Scene(desc)
    want = one.render(params)
    one.close()
    film = np.zeros((H, W, 3), dtype=np.float32)
    scenes = [dev.Scene(desc) for _ in range(world)]

    def work(r):
        tiles, n = host.tiles(W, H, 32, r, world)
        scenes[r].render(desc.render_params(spp=8, seed=6), tiles, n, out=film)
    ts = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert np.array_equal(film, want)
    [s.close() for s in scenes]


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
