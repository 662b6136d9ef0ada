"""ctypes binding of liblumilly_hip.so (include/lumilly_hip.h): the HIP render path.

There is NO fallback: if the extension is missing or no GPU is visible, construction raises.
Counterpart of the reference's `Scene` as the render loop sees it (scene.rs:20,34 behind
main.rs:70-132): `Scene(description).render(params, tiles)` returns the film.
"""
import ctypes as C
import os

import numpy as np

from . import abi
from .host import LumillyError

_HERE = os.path.dirname(os.path.abspath(__file__))
# LR_HIP_LIB=<path> loads another build of the SAME library (diagnostic variants under build/: `make -C csrc diag stamp`);
# there is still no fallback -- a missing file raises.
_LIB_PATH = os.environ.get("LR_HIP_LIB") or os.path.join(_HERE, "liblumilly_hip.so")
_lib = None


KNOBS_LIB_PATH = os.path.join(_HERE, "liblumilly_hip_knobs.so")      # `make -C lumillyrender_amd/csrc knobs`: the diagnostic knobs compiled in (csrc/lr_knobs.h)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    _lib = load_library(_LIB_PATH)
    return _lib


def load_library(path):
    """dlopen one build of the library and declare its entry points.  `lib()` is the process's product library; the tests load the
    knob build beside it (tests/conftest.py) -- the product library reads no LR_* variable that changes what it runs."""
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: the HIP extension was not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C lumillyrender_amd/csrc`). "
            "There is no CPU fallback for the render path.")
    l = C.CDLL(path)
    vp = C.c_void_p
    fp = C.POINTER(C.c_float)
    l.lr_last_error.restype = C.c_char_p
    l.lr_build_info.restype = C.c_char_p
    l.lr_device_count.restype = C.c_int
    l.lr_scene_create.argtypes = [C.c_int, C.POINTER(abi.LrSceneDesc), C.POINTER(vp)]
    l.lr_scene_destroy.argtypes = [vp]
    l.lr_render.argtypes = [vp, C.POINTER(abi.LrRenderParams), C.POINTER(abi.LrTile), C.c_int, fp, C.c_size_t]
    l.lr_render_device.argtypes = [vp, C.POINTER(abi.LrRenderParams), C.POINTER(abi.LrTile), C.c_int, C.POINTER(vp)]
    l.lr_get_stats.argtypes = [vp, C.POINTER(abi.LrStats)]
    l.lr_film_quantize.argtypes = [vp, C.c_int, C.c_float, C.POINTER(C.c_uint8), C.c_size_t]
    l.lr_selftest_math.argtypes = [C.c_int, C.c_int, fp, fp, fp, C.c_int]
    up = C.POINTER(C.c_uint32)
    l.lr_selftest_rng.argtypes = [C.c_int, C.c_uint32, up, up, up, fp, C.c_int]
    l.lr_selftest_intersect.argtypes = [vp, C.c_int, fp, fp, C.POINTER(C.c_int32), fp]
    l.lr_selftest_brute.argtypes = [vp, C.c_int, fp, fp, C.POINTER(C.c_int32), fp]
    if hasattr(l, "lr_selftest_brute_own_box"):            # round 6; older builds (tools/ab4.py baselines) lack it
        l.lr_selftest_brute_own_box.argtypes = [vp, C.c_int, fp, fp, C.POINTER(C.c_int32), fp]
    l.lr_selftest_emitter_pick.argtypes = [vp, C.c_int, fp, C.POINTER(C.c_int32)]
    l.lr_selftest_sky.argtypes = [vp, C.c_int, fp, fp]
    if hasattr(l, "lr_selftest_material"):                 # diagnostics entry points of round 4; older builds (tools/ab4.py baselines) lack them
        l.lr_selftest_sky_texel_bytes.argtypes = [vp]
        l.lr_selftest_tree_info.argtypes = [vp, C.POINTER(C.c_int32)]
        l.lr_selftest_material.argtypes = [C.c_int, C.POINTER(abi.LrMaterial), C.c_int, fp, fp]
        l.lr_selftest_camera.argtypes = [vp, C.c_int, C.POINTER(C.c_int32), fp, fp]
        l.lr_selftest_emission_sample.argtypes = [vp, C.c_int, fp, fp]
    if hasattr(l, "lr_selftest_rcp"):                      # diagnostics entry point; older builds (tools/sweep.sh) lack it
        l.lr_selftest_rcp.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]
    return l


def build_info():
    """lr_build_info(): target, float mode, ABI version and the content hash of the sources + flags the library was built from."""
    return lib().lr_build_info().decode()


def build_id():
    """The `build=` field of lr_build_info (tools/build_id.py): what ties a committed profile to the library that was profiled."""
    info = build_info()
    return info.split("build=", 1)[1].split()[0] if "build=" in info else "unknown"


def _check(rc):
    if rc < 0:
        raise LumillyError(rc, lib().lr_last_error().decode("utf-8", "replace"))
    return rc


def device_count():
    return lib().lr_device_count()


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class Scene:
    """Device-resident scene (BVH, primitives, materials, emitters, sky) on one GPU."""

    def __init__(self, description, device=0, device_bvh=False):
        """device_bvh=True drops the host SAH tree from the description so that lr_scene_create builds an
        LBVH on the GPU (same images: the tree only prunes)."""
        self._h = C.c_void_p()
        self.description = description            # keeps the host arrays alive during create
        if device_count() <= 0:
            raise RuntimeError("lumilly_hip: no HIP device visible (the render path has no CPU fallback)")
        ptr = description.desc_ptr
        if device_bvh:
            d = abi.LrSceneDesc.from_buffer_copy(description.desc)
            d.n_bvh_nodes, d.bvh_max_depth = 0, 0
            d.bvh_nodes = C.POINTER(abi.LrBvhNode)()
            d.bvh_prim_order = C.POINTER(C.c_int32)()
            ptr = C.pointer(d)
        _check(lib().lr_scene_create(device, ptr, C.byref(self._h)))
        d = description.desc
        self.width, self.height = int(d.camera.resolution[0]), int(d.camera.resolution[1])
        self.device = device

    def close(self):
        if self._h:
            lib().lr_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _tile_args(self, tiles, n_tiles):
        if tiles is None:
            tiles = (abi.LrTile * 1)()
            tiles[0].x0, tiles[0].y0, tiles[0].w, tiles[0].h = 0, 0, self.width, self.height
            return tiles, 1
        if n_tiles is None:
            n_tiles = len(tiles)
        if n_tiles < 0 or n_tiles > len(tiles):
            raise ValueError(f"n_tiles = {n_tiles} but the tile array holds {len(tiles)}")
        return tiles, n_tiles

    def _check_film(self, out):
        """The native side writes (H, W, 3) f32 rows of the film size captured at scene creation: anything else
        (another dtype, a strided view, a film resized after Scene()) would be written out of bounds."""
        if not isinstance(out, np.ndarray) or out.dtype != np.float32:
            raise ValueError("film must be a float32 numpy array")
        if out.shape != (self.height, self.width, 3):
            raise ValueError(f"film shape {out.shape} does not match the scene's film ({self.height}, {self.width}, 3)")
        if not out.flags.c_contiguous or not out.flags.writeable:
            raise ValueError("film must be C-contiguous and writable")

    def render(self, params, tiles=None, n_tiles=None, out=None):
        """Render `tiles` (default: the whole film) into an (H, W, 3) f32 array."""
        tiles, n_tiles = self._tile_args(tiles, n_tiles)
        if out is None:
            out = np.zeros((self.height, self.width, 3), dtype=np.float32)
        self._check_film(out)
        _check(lib().lr_render(self._h, C.byref(params), tiles, n_tiles, _fptr(out), self.width * 3))
        return out

    def render_device(self, params, tiles=None, n_tiles=None):
        """Render and leave the film in HBM; returns the device pointer (int)."""
        tiles, n_tiles = self._tile_args(tiles, n_tiles)
        p = C.c_void_p()
        _check(lib().lr_render_device(self._h, C.byref(params), tiles, n_tiles, C.byref(p)))
        return p.value

    def quantize(self, mode="rgb8", gamma=2.2):
        """Film output stage on the device for the last render: (H, W, 3) uint8 with the reference's
        gamma/truncation (main.rs:171-173) or (H, W, 4) Radiance RGBE bytes."""
        if mode not in ("rgb8", "rgbe"):
            raise ValueError("mode must be 'rgb8' or 'rgbe'")
        bpp = 3 if mode == "rgb8" else 4
        out = np.empty((self.height, self.width, bpp), dtype=np.uint8)
        _check(lib().lr_film_quantize(self._h, 0 if mode == "rgb8" else 1, gamma, out.ctypes.data_as(C.POINTER(C.c_uint8)), self.width * bpp))
        return out

    def stats(self):
        s = abi.LrStats()
        _check(lib().lr_get_stats(self._h, C.byref(s)))
        return s

    @staticmethod
    def _rays(origins, dirs):
        o = np.ascontiguousarray(origins, dtype=np.float32)
        d = np.ascontiguousarray(dirs, dtype=np.float32)
        if o.ndim != 2 or o.shape[1] != 3 or d.shape != o.shape:
            raise ValueError(f"origins / dirs must both be (n, 3) arrays, got {o.shape} and {d.shape}")
        return o, d

    def intersect(self, origins, dirs, brute=False):
        """Closest hit per ray: through the traversal path lr_render uses (flat loop or tree); with brute=True against every
        primitive behind its own exact box (the definition, bvh.rs:20-25 + 131-141, evaluated per primitive); with
        brute="all" against every primitive without any box (the definition of rounds 1-5).
        Returns (primitive index or -1, distance)."""
        o, d = self._rays(origins, dirs)
        n = o.shape[0]
        prim = np.empty(n, dtype=np.int32)
        t = np.empty(n, dtype=np.float32)
        fn = lib().lr_selftest_brute if brute == "all" else (lib().lr_selftest_brute_own_box if brute else lib().lr_selftest_intersect)
        _check(fn(self._h, n, _fptr(o), _fptr(d), prim.ctypes.data_as(C.POINTER(C.c_int32)), _fptr(t)))
        return prim, t

    def sky(self, dirs):
        """sky.rs on the device: radiance seen along each unit direction, (n, 3)."""
        d = np.ascontiguousarray(dirs, dtype=np.float32)
        if d.ndim != 2 or d.shape[1] != 3:
            raise ValueError("dirs must be (n, 3)")
        out = np.empty_like(d)
        _check(lib().lr_selftest_sky(self._h, d.shape[0], _fptr(d), _fptr(out)))
        return out

    def camera_samples(self, xy, xi4):
        """camera.rs sample() of the scene's camera on the device: (n, 8) = origin, direction, geometry term, 0."""
        p = np.ascontiguousarray(xy, dtype=np.int32).reshape(-1, 2)
        x = np.ascontiguousarray(xi4, dtype=np.float32).reshape(-1, 4)
        out = np.empty((p.shape[0], 8), dtype=np.float32)
        _check(lib().lr_selftest_camera(self._h, p.shape[0], p.ctypes.data_as(C.POINTER(C.c_int32)), _fptr(x), _fptr(out)))
        return out

    def emission_sample(self, xi4):
        """objects.rs:37-51 in full on the device: (n, 4) = sampled point, pdf."""
        x = np.ascontiguousarray(xi4, dtype=np.float32).reshape(-1, 4)
        out = np.empty_like(x)
        _check(lib().lr_selftest_emission_sample(self._h, x.shape[0], _fptr(x), _fptr(out)))
        return out

    def tree_info(self):
        """{nodes, nodes_without_distance_culling (culling slack >= the distance itself), sliver_triangles, stack_need} of the scene's 4-wide tree."""
        out = (C.c_int32 * 4)()
        _check(lib().lr_selftest_tree_info(self._h, out))
        return {"nodes": out[0], "nodes_without_distance_culling": out[1], "sliver_triangles": out[2], "stack_need": out[3]}

    def sky_texel_bytes(self):
        """How the IBL map is stored on the device: 4 (RGBE words, exact decode), 16 (float4) or 0 (no map)."""
        return _check(lib().lr_selftest_sky_texel_bytes(self._h))

    def emitter_pick(self, xi):
        """objects.rs:37-51 on the device: emitter index chosen for each uniform draw."""
        x = np.ascontiguousarray(xi, dtype=np.float32).reshape(-1)
        k = np.empty(x.size, dtype=np.int32)
        _check(lib().lr_selftest_emitter_pick(self._h, x.size, _fptr(x), k.ctypes.data_as(C.POINTER(C.c_int32))))
        return k


def selftest_material(material, in13, device=0):
    """material/*.rs on the device for one abi.LrMaterial: in13 (n, 13) = out_, normal, position, xi[3], fly distance ->
    (n, 10) = sampled in_, pdf, brdf(out_, in_), coef."""
    a = np.ascontiguousarray(in13, dtype=np.float32).reshape(-1, 13)
    out = np.empty((a.shape[0], 10), dtype=np.float32)
    _check(lib().lr_selftest_material(device, C.byref(material), a.shape[0], _fptr(a), _fptr(out)))
    return out


def selftest_math(fn, a, b=None, device=0):
    a = np.ascontiguousarray(a, dtype=np.float32)
    out = np.empty_like(a)
    bp = None
    if b is not None:
        b = np.ascontiguousarray(b, dtype=np.float32)
        if b.shape != a.shape:
            raise ValueError("selftest_math: a and b differ in shape")
        bp = _fptr(b)
    _check(lib().lr_selftest_math(device, fn, _fptr(a), bp, _fptr(out), a.size))
    return out


def selftest_rng(seed, pixel, sample, block, device=0):
    pixel = np.ascontiguousarray(pixel, dtype=np.uint32)
    sample = np.ascontiguousarray(sample, dtype=np.uint32)
    block = np.ascontiguousarray(block, dtype=np.uint32)
    out = np.empty((pixel.size, 4), dtype=np.float32)
    up = C.POINTER(C.c_uint32)
    _check(lib().lr_selftest_rng(device, seed, pixel.ctypes.data_as(up), sample.ctypes.data_as(up), block.ctypes.data_as(up), _fptr(out), pixel.size))
    return out


def selftest_rcp(lo_exp, hi_exp, device=0):
    """Exhaustive check of the device's five-instruction exact reciprocal against IEEE 1/d over every float whose
    biased exponent lies in [lo_exp, hi_exp]: returns (mismatches of the 2-step form, of the 3-step form, example bits)."""
    out = (C.c_uint64 * 4)()
    _check(lib().lr_selftest_rcp(device, lo_exp, hi_exp, out))
    return int(out[0]), int(out[1]), (int(out[2]), int(out[3]))


def stats_dict(s):
    d = {k: int(getattr(s, k)) for k in ("samples", "segments", "shadow_rays", "node_visits", "prim_tests", "shadow_node_visits", "shadow_prim_tests", "sky_fetches", "iterations", "path_slots", "pipeline")}
    d["render_ms"] = float(s.render_ms)
    d["upload_ms"] = float(s.upload_ms)
    d["bvh_build_ms"] = float(s.bvh_build_ms)
    d["kernels"] = {
        abi.LR_KERNEL_NAMES[k]: {"launches": int(s.kernel_launches[k]), "timed": int(s.kernel_timed[k]), "ms": float(s.kernel_ms[k])}
        for k in range(abi.LR_K_COUNT)
    }
    return d
