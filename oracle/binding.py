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
                ("prim_tests", C.c_uint64), ("sky_fetches", C.c_uint64), ("seconds", C.c_double)]


_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None
fp = C.POINTER(C.c_float)


def build():
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    l = C.CDLL(_LIB_PATH)
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
    l.lr_oracle_sky_radiance.argtypes = [C.POINTER(abi.LrSceneDesc), fp, fp]
    l.lr_oracle_sky_radiance.restype = None
    for n in ("sin", "cos", "acos", "exp"):
        f = getattr(l, "lr_oracle_" + n)
        f.argtypes, f.restype = [C.c_float], C.c_float
    for n in ("atan2", "pow", "fmod_pos"):
        f = getattr(l, "lr_oracle_" + n)
        f.argtypes, f.restype = [C.c_float, C.c_float], C.c_float
    _lib = l
    return l


def f3(v):
    return (C.c_float * len(v))(*[float(x) for x in v])


BRUTE, BVH = 0, 1


def render(description, params, tiles=None, n_tiles=None, threads=0, mode=BRUTE, pad=0.0, with_stats=False):
    """Render with the CPU oracle.  mode BRUTE = the closest-hit definition; BVH with pad == 0 is the
    reference's literal tree + candidate traversal (bvh.rs / aabb.rs)."""
    d = description.desc
    w, h = int(d.camera.resolution[0]), int(d.camera.resolution[1])
    if tiles is None:
        tiles = (abi.LrTile * 1)()
        tiles[0].x0, tiles[0].y0, tiles[0].w, tiles[0].h = 0, 0, w, h
        n_tiles = 1
    img = np.zeros((h, w, 3), dtype=np.float32)
    st = LrOracleStats()
    rc = lib().lr_oracle_render(description.desc_ptr, C.byref(params), tiles, n_tiles, img.ctypes.data_as(fp), w * 3,
                                threads, mode, pad, C.byref(st))
    if rc != 0:
        raise RuntimeError(f"lr_oracle_render failed: {rc}")
    return (img, st) if with_stats else img


def intersect(description, origins, dirs, mode=BRUTE, pad=0.0):
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
