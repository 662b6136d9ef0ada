"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

May be imported by tests/, by __graft_entry__.smoke() and by bench.py's cpu_baseline leg, and by
nothing under lumillyrender_amd/.  See oracle/lr_oracle.cpp for what the oracle restates and what
pins it.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))
from lumillyrender_amd import abi  # noqa: E402  (POD struct mirrors only)



class LrOracleStats(C.Structure):
    """lr_oracle.cpp: struct LrOracleStats"""
    _fields_ = [("samples", C.c_uint64), ("segments", C.c_uint64), ("shadow_rays", C.c_uint64), ("node_visits", C.c_uint64),
                ("prim_tests", C.c_uint64), ("sky_fetches", C.c_uint64), ("seconds", C.c_double),
                ("tie_flips", C.c_uint64), ("order_dependent", C.c_uint64)]


# LR_ORACLE_LIB=<path>: the sanitizer build of the checker (`make -C oracle asan`), used for both roles
_LIB_PATH = os.environ.get("LR_ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")
_FAST_PATH = os.environ.get("LR_ORACLE_LIB") or os.path.join(_HERE, "liboracle_fast.so")      # same source, -O3 -mavx2 (the cpu_baseline build of BASELINE.md section 3)
_libs = {}
fp = C.POINTER(C.c_float)


def build():
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def lib(fast=False):
    """fast=False: the parity build (-O2).  fast=True: the -O3 -mavx2 build timed as the CPU baseline; the two
    must give identical bits (tests/test_oracle_properties.py checks it)."""
    path = _FAST_PATH if fast else _LIB_PATH
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        build()
    l = C.CDLL(path)
    l.lr_oracle_render.argtypes = [C.POINTER(abi.LrSceneDesc), C.POINTER(abi.LrRenderParams), C.POINTER(abi.LrTile), C.c_int,
                                   fp, C.c_size_t, C.c_int, C.c_int, C.c_float, C.POINTER(LrOracleStats)]
    l.lr_oracle_triangle_intersect.argtypes = [fp, fp, fp, C.c_int, fp]
    l.lr_oracle_sphere_intersect.argtypes = [fp, C.c_float, fp, fp, fp]
    l.lr_oracle_reflect.argtypes = [fp, fp, fp]
    l.lr_oracle_reflect.restype = None
    l.lr_oracle_refract.argtypes = [fp, fp, C.c_float, fp]
    l.lr_oracle_orthonormal_basis.argtypes = [fp, fp, fp]
    l.lr_oracle_orthonormal_basis.restype = None
    mp = C.POINTER(abi.LrMaterial)
    l.lr_oracle_material_brdf.argtypes = [mp, fp, fp, fp, fp, fp]
    l.lr_oracle_material_brdf.restype = None
    l.lr_oracle_material_sample.argtypes = [mp, fp, fp, fp, fp, fp]
    l.lr_oracle_material_sample.restype = None
    l.lr_oracle_material_weight.argtypes = [mp]
    l.lr_oracle_material_weight.restype = C.c_float
    l.lr_oracle_material_coef.argtypes = [mp, fp, fp, C.c_float, fp]
    l.lr_oracle_material_coef.restype = None
    l.lr_oracle_ior_pair.argtypes = [mp, fp, fp, fp]
    l.lr_oracle_ior_pair.restype = None
    l.lr_oracle_fresnel.argtypes = [C.c_float, C.c_float, fp, fp, fp]
    l.lr_oracle_fresnel.restype = C.c_float
    l.lr_oracle_checker.argtypes = [C.c_float, C.c_float, fp]
    l.lr_oracle_checker.restype = None
    l.lr_oracle_russian_roulette.argtypes = [C.c_float, C.c_int, C.c_int, C.c_int]
    l.lr_oracle_russian_roulette.restype = C.c_float
    l.lr_oracle_rng_block.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, fp]
    l.lr_oracle_rng_block.restype = None
    l.lr_oracle_camera_sample.argtypes = [C.POINTER(abi.LrCamera), C.c_int, C.c_int, fp, fp]
    l.lr_oracle_camera_sample.restype = None
    l.lr_oracle_prim_sample.argtypes = [C.POINTER(abi.LrPrimitive), C.c_float, C.c_float, fp]
    l.lr_oracle_prim_sample.restype = None
    l.lr_oracle_intersect_batch.argtypes = [C.POINTER(abi.LrSceneDesc), C.c_int, C.c_float, C.c_int, fp, fp, C.POINTER(C.c_int32), fp]
    l.lr_oracle_bvh_leaf_order.argtypes = [C.POINTER(abi.LrSceneDesc), C.POINTER(C.c_int32)]
    l.lr_oracle_sky_batch.argtypes = [C.POINTER(abi.LrSceneDesc), C.c_int, fp, fp]
    l.lr_oracle_sky_batch.restype = None
    l.lr_oracle_emitter_pick.argtypes = [C.POINTER(abi.LrSceneDesc), C.c_int, fp, C.POINTER(C.c_int32)]
    l.lr_oracle_math_batch.argtypes = [C.c_int, fp, fp, fp, C.c_int]
    l.lr_oracle_emission_sample.argtypes = [C.POINTER(abi.LrSceneDesc), C.c_int, fp, fp]
    l.lr_oracle_aabb_is_intersect.argtypes = [fp, fp, fp]
    l.lr_oracle_sky_radiance.argtypes = [C.POINTER(abi.LrSceneDesc), fp, fp]
    l.lr_oracle_sky_radiance.restype = None
    for n in ("sin", "cos", "acos", "exp"):
        f = getattr(l, "lr_oracle_" + n)
        f.argtypes, f.restype = [C.c_float], C.c_float
    for n in ("atan2", "pow", "fmod_pos"):
        f = getattr(l, "lr_oracle_" + n)
        f.argtypes, f.restype = [C.c_float, C.c_float], C.c_float
    _libs[path] = l
    return l


def f3(v):
    return (C.c_float * len(v))(*[float(x) for x in v])


BRUTE, BVH, BVH_ORDERED, OWNBOX, OWNBOX_TREE, OWNBOX_ORDERED, BVH_AUDIT, OWNBOX_INDEX, OWNBOX_TREE_INDEX = 0, 1, 2, 3, 4, 5, 6, 7, 8
# OWNBOX (every primitive behind its own exact box, bvh.rs:20-25 + aabb.rs:74-92; exact ties to the first primitive of the reference's
# candidate order) is what the HIP path implements; BVH with pad == 0 is the reference line by line and gives the same result on
# every ray (OWNBOX_TREE is its alias: fast on the 100k-triangle scenes); the _INDEX forms break exact ties by primitive index, which
# is what a device-built tree gives.


def render(description, params, tiles=None, n_tiles=None, threads=0, mode=OWNBOX, pad=0.0, with_stats=False, fast=False):
    """Render with the CPU oracle.  mode OWNBOX = the closest-hit definition (every primitive behind its own exact box);
    BVH with pad == 0 is the reference's literal tree + candidate traversal (bvh.rs / aabb.rs); OWNBOX_ORDERED needs pad > 0."""
    d = description.desc
    w, h = int(d.camera.resolution[0]), int(d.camera.resolution[1])
    if tiles is None:
        tiles = (abi.LrTile * 1)()
        tiles[0].x0, tiles[0].y0, tiles[0].w, tiles[0].h = 0, 0, w, h
        n_tiles = 1
    img = np.zeros((h, w, 3), dtype=np.float32)
    st = LrOracleStats()
    rc = lib(fast).lr_oracle_render(description.desc_ptr, C.byref(params), tiles, n_tiles, img.ctypes.data_as(fp), w * 3,
                                threads, mode, pad, C.byref(st))
    if rc != 0:
        raise RuntimeError(f"lr_oracle_render failed: {rc}")
    return (img, st) if with_stats else img


def intersect(description, origins, dirs, mode=OWNBOX, pad=0.0):
    o = np.ascontiguousarray(origins, dtype=np.float32)
    d = np.ascontiguousarray(dirs, dtype=np.float32)
    n = o.shape[0]
    prim = np.empty(n, dtype=np.int32)
    t = np.empty(n, dtype=np.float32)
    rc = lib().lr_oracle_intersect_batch(description.desc_ptr, mode, pad, n, o.ctypes.data_as(fp), d.ctypes.data_as(fp),
                                         prim.ctypes.data_as(C.POINTER(C.c_int32)), t.ctypes.data_as(fp))
    if rc != 0:
        raise RuntimeError(f"lr_oracle_intersect_batch failed: {rc}")
    return prim, t


def bvh_leaf_order(description):
    """Primitive indices in the candidate order of the reference's tree (depth-first, left first; bvh.rs:38-45)."""
    n = int(description.desc.n_prims)
    out = np.empty(n, dtype=np.int32)
    k = lib().lr_oracle_bvh_leaf_order(description.desc_ptr, out.ctypes.data_as(C.POINTER(C.c_int32)))
    if k != n:
        raise RuntimeError(f"lr_oracle_bvh_leaf_order: {k}")
    return out


def sky_batch(description, dirs):
    d = np.ascontiguousarray(dirs, dtype=np.float32)
    out = np.empty_like(d)
    lib().lr_oracle_sky_batch(description.desc_ptr, d.shape[0], d.ctypes.data_as(fp), out.ctypes.data_as(fp))
    return out


def emitter_pick(description, xi):
    """objects.rs:37-51: (emitter index per draw, number of emitters)."""
    x = np.ascontiguousarray(xi, dtype=np.float32).reshape(-1)
    k = np.empty(x.size, dtype=np.int32)
    n = lib().lr_oracle_emitter_pick(description.desc_ptr, x.size, x.ctypes.data_as(fp), k.ctypes.data_as(C.POINTER(C.c_int32)))
    if n < 0:
        raise RuntimeError("lr_oracle_emitter_pick failed")
    return k, n


_MATH_FN = {"sin": 0, "cos": 1, "acos": 2, "atan2": 3, "pow": 4, "exp": 5, "fmod_pos": 6}


def emission_sample(description, xi4):
    """Objects::sample_emission (objects.rs:37-51) for draws (n, 4) = (-, pick, u, v): (n, 4) = point.xyz, pdf."""
    x = np.ascontiguousarray(xi4, dtype=np.float32).reshape(-1, 4)
    out = np.empty_like(x)
    if lib().lr_oracle_emission_sample(description.desc_ptr, x.shape[0], x.ctypes.data_as(fp), out.ctypes.data_as(fp)) != 0:
        raise RuntimeError("lr_oracle_emission_sample failed")
    return out


def math_batch(name, xs, ys=None):
    """The oracle's deterministic math spec over whole arrays (one C call; for the exhaustive-ish sweeps)."""
    a = np.ascontiguousarray(xs, dtype=np.float32)
    out = np.empty_like(a)
    bp = None
    if ys is not None:
        b = np.ascontiguousarray(ys, dtype=np.float32)
        bp = b.ctypes.data_as(fp)
    rc = lib().lr_oracle_math_batch(_MATH_FN[name], a.ctypes.data_as(fp), bp, out.ctypes.data_as(fp), a.size)
    assert rc == 0
    return out


def math1(name, xs):
    f = getattr(lib(), "lr_oracle_" + name)
    return np.array([f(float(x)) for x in np.asarray(xs, dtype=np.float32)], dtype=np.float32)


def math2(name, xs, ys):
    f = getattr(lib(), "lr_oracle_" + name)
    return np.array([f(float(x), float(y)) for x, y in zip(np.asarray(xs, dtype=np.float32), np.asarray(ys, dtype=np.float32))], dtype=np.float32)


def rng_block(seed, pixel, sample, block):
    out = (C.c_float * 4)()
    lib().lr_oracle_rng_block(seed, pixel, sample, block, out)
    return np.array(out[:], dtype=np.float32)
