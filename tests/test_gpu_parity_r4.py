"""GPU parity, round 4: the IBL map stored as exact-decoding RGBE words (sky.rs:45-48,57-78), the pinned size of the
grazing-vertex residual of distance culling (bvh.rs:131-141 vs triangle.rs:75), the golden fixtures consumed WITHOUT the
oracle (tests/golden/), and an 8-rank rehearsal of the multi-GPU bench path on one GPU."""
import ctypes as C
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


def _directions(rng, n):
    d = rng.standard_normal((n, 3))
    axes = np.array([[0, 1, 0], [0, -1, 0], [1, 0, 0], [-1, 0, 0], [0, 0, 1], [0, 0, -1]], dtype=np.float64)
    d = np.concatenate([d, axes])
    return (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)


# ---- IBL texels as RGBE words --------------------------------------------------------------------------------------

def test_ibl_map_is_stored_as_rgbe_and_decodes_to_the_same_bits(dev, oracle, monkeypatch):
    """A map that was loaded from an .hdr file holds Radiance values c * 2^(e - 136) (the `image` crate's decode behind
    sky.rs:45-48); lr_scene_create re-encodes every texel, checks the device's decode of the whole map against the caller's
    floats and then keeps 4 B per texel instead of 16.  Same texel (sky.rs:57-78), same f32 bits: lookups equal the oracle's
    and the float4 build's bit for bit, films and counters of the two storage forms are identical."""
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
