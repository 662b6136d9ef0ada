"""ctypes binding of liblumilly_host.so (include/lumilly_host.h): scene loading, BVH build, tile
queue and film output.  Mirrors the reference's `Description` (description.rs:27-82): a loaded
scene exposes `.config` style accessors, `.camera`, and the flat `LrSceneDesc` the device consumes.
"""
import ctypes as C
import json
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(_HERE)
ASSET_ROOT = os.path.join(REPO_ROOT, "assets")
# LR_HOST_LIB=<path>: another build of the SAME library (the sanitizer build of `make -C host asan`, tests/test_sanitizers.py)
_LIB_PATH = os.environ.get("LR_HOST_LIB") or os.path.join(_HERE, "liblumilly_host.so")


class LumillyError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"[{code}] {msg}")
        self.code = code


def _load():
    if not os.path.exists(_LIB_PATH):
        raise ImportError(
            f"{_LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C lumillyrender_amd/host`)")
    lib = C.CDLL(_LIB_PATH)
    vp = C.c_void_p
    lib.lr_host_last_error.restype = C.c_char_p
    lib.lr_host_load_scene.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(vp)]
    lib.lr_host_load_scene_string.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(vp)]
    lib.lr_host_scene_free.argtypes = [vp]
    lib.lr_host_scene_free.restype = None
    lib.lr_host_scene_desc.argtypes = [vp]
    lib.lr_host_scene_desc.restype = C.POINTER(abi.LrSceneDesc)
    lib.lr_host_scene_renderer.argtypes = [vp, C.POINTER(abi.LrRendererConfig)]
    lib.lr_host_scene_film.argtypes = [vp, C.POINTER(abi.LrFilmConfig)]
    lib.lr_host_scene_set_resolution.argtypes = [vp, C.c_int, C.c_int]
    lib.lr_host_scene_bvh_info.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double)]
    lib.lr_host_scene_dump_json.argtypes = [vp, C.c_int, C.POINTER(vp)]
    lib.lr_host_build_bvh.argtypes = [C.POINTER(abi.LrPrimitive), C.c_int, C.c_int, C.POINTER(C.c_float),
                                      C.POINTER(C.POINTER(abi.LrBvhNode)), C.POINTER(C.c_int),
                                      C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.c_int)]
    lib.lr_host_free.argtypes = [vp]
    lib.lr_host_free.restype = None
    fp = C.POINTER(C.c_float)
    lib.lr_host_save_png.argtypes = [C.c_char_p, fp, C.c_int, C.c_int, C.c_size_t, C.c_float]
    lib.lr_host_save_hdr.argtypes = [C.c_char_p, fp, C.c_int, C.c_int, C.c_size_t]
    bp = C.POINTER(C.c_uint8)
    lib.lr_host_write_png_rgb8.argtypes = [C.c_char_p, bp, C.c_int, C.c_int, C.c_size_t]
    lib.lr_host_write_hdr_rgbe.argtypes = [C.c_char_p, bp, C.c_int, C.c_int, C.c_size_t]
    lib.lr_host_to_color.argtypes = [fp, C.c_size_t, C.c_float, C.POINTER(C.c_uint8)]
    lib.lr_host_load_hdr.argtypes = [C.c_char_p, C.POINTER(fp), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.lr_host_tiles.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(abi.LrTile), C.c_int]
    lib.lr_host_tile_rank.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.lr_host_tile_stride.argtypes = [C.c_int]
    lib.lr_host_default_tile.argtypes = []
    lib.lr_host_shm_barrier.argtypes = [C.c_void_p, C.c_int, C.c_double]
    lib.lr_host_sizeof.argtypes = [C.c_char_p]
    lib.lr_host_sizeof.restype = C.c_size_t
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def _check(rc):
    if rc < 0:
        raise LumillyError(rc, lib().lr_host_last_error().decode("utf-8", "replace"))
    return rc


class Description:
    """Counterpart of the reference's `Description::new(path)` (description.rs:32-44)."""

    def __init__(self, path=None, text=None, asset_root=ASSET_ROOT):
        self._h = C.c_void_p()
        root = asset_root.encode() if asset_root else None
        if text is not None:
            _check(lib().lr_host_load_scene_string(text.encode(), root, C.byref(self._h)))
        else:
            _check(lib().lr_host_load_scene(os.fspath(path).encode(), root, C.byref(self._h)))

    def close(self):
        if self._h:
            lib().lr_host_scene_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def desc_ptr(self):
        return lib().lr_host_scene_desc(self._h)

    @property
    def desc(self):
        return self.desc_ptr.contents

    @property
    def renderer(self):
        r = abi.LrRendererConfig()
        _check(lib().lr_host_scene_renderer(self._h, C.byref(r)))
        return r

    @property
    def film(self):
        f = abi.LrFilmConfig()
        _check(lib().lr_host_scene_film(self._h, C.byref(f)))
        return f

    @property
    def resolution(self):
        f = self.film
        return int(f.resolution[0]), int(f.resolution[1])

    def set_resolution(self, width, height):
        _check(lib().lr_host_scene_set_resolution(self._h, int(width), int(height)))

    def bvh_info(self):
        s, n, d, c = C.c_double(), C.c_int(), C.c_int(), C.c_double()
        _check(lib().lr_host_scene_bvh_info(self._h, C.byref(s), C.byref(n), C.byref(d), C.byref(c)))
        return {"seconds": s.value, "nodes": n.value, "max_depth": d.value, "sah_cost": c.value}

    def dump(self, max_prims=-1):
        p = C.c_void_p()
        _check(lib().lr_host_scene_dump_json(self._h, max_prims, C.byref(p)))
        try:
            return json.loads(C.string_at(p).decode())
        finally:
            lib().lr_host_free(p)

    def render_params(self, spp=None, seed=0, integrator=None, flags=0, path_slots=0):
        """[renderer] table -> LrRenderParams (description.rs:74-79, main.rs:66)."""
        r = self.renderer
        p = abi.LrRenderParams()
        p.integrator = r.integrator if integrator is None else integrator
        p.spp = r.samples if spp is None else spp
        p.seed = seed
        p.depth = r.depth
        p.depth_limit = r.depth_limit
        p.no_direct_emitter = r.no_direct_emitter
        p.path_slots = path_slots
        p.flags = flags
        return p


def shm_barrier(state_address, world, timeout_s=120.0):
    """Barrier of `world` processes on two uint32 words of shared memory at `state_address` (lumilly_host.h)."""
    _check(lib().lr_host_shm_barrier(C.c_void_p(state_address), int(world), float(timeout_s)))


def default_tile():
    return int(lib().lr_host_default_tile())


def tile_rank(i, j, world):
    return int(lib().lr_host_tile_rank(i, j, world))


def tile_split_name():
    return "tile (i, j) -> rank (i + k j) mod world, k = lr_host_tile_stride(world) (8: %d, 4: %d, 2: %d); default tile %d px" % (
        lib().lr_host_tile_stride(8), lib().lr_host_tile_stride(4), lib().lr_host_tile_stride(2), default_tile())


def tiles(width, height, tile=0, rank=0, world=1):
    """Pixel tile queue shard of `rank`: block (i, j) of the tile grid goes to rank (i + k j) mod world (lumilly_host.h);
    tile = 0: the library's default size."""
    if tile <= 0:
        tile = default_tile()
    n = _check(lib().lr_host_tiles(width, height, tile, rank, world, None, 0))
    arr = (abi.LrTile * max(n, 1))()
    _check(lib().lr_host_tiles(width, height, tile, rank, world, arr, n))
    return arr, n


def full_tile(width, height):
    arr = (abi.LrTile * 1)()
    arr[0].x0, arr[0].y0, arr[0].w, arr[0].h = 0, 0, width, height
    return arr, 1


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def save_png(path, rgb, gamma=2.2):
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    h, w, _ = rgb.shape
    _check(lib().lr_host_save_png(os.fspath(path).encode(), _fptr(rgb), w, h, w * 3, gamma))


def save_hdr(path, rgb):
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    h, w, _ = rgb.shape
    _check(lib().lr_host_save_hdr(os.fspath(path).encode(), _fptr(rgb), w, h, w * 3))


def write_png_rgb8(path, rgb8):
    rgb8 = np.ascontiguousarray(rgb8, dtype=np.uint8)
    h, w, _ = rgb8.shape
    _check(lib().lr_host_write_png_rgb8(os.fspath(path).encode(), rgb8.ctypes.data_as(C.POINTER(C.c_uint8)), w, h, w * 3))


def write_hdr_rgbe(path, rgbe):
    rgbe = np.ascontiguousarray(rgbe, dtype=np.uint8)
    h, w, _ = rgbe.shape
    _check(lib().lr_host_write_hdr_rgbe(os.fspath(path).encode(), rgbe.ctypes.data_as(C.POINTER(C.c_uint8)), w, h, w * 4))


def load_hdr(path):
    p, w, h = C.POINTER(C.c_float)(), C.c_int(), C.c_int()
    _check(lib().lr_host_load_hdr(os.fspath(path).encode(), C.byref(p), C.byref(w), C.byref(h)))
    try:
        return np.ctypeslib.as_array(p, shape=(h.value, w.value, 3)).copy()
    finally:
        lib().lr_host_free(p)


def to_color(rgb, gamma=2.2):
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    out = np.empty(rgb.shape, dtype=np.uint8)
    _check(lib().lr_host_to_color(_fptr(rgb), rgb.size, gamma, out.ctypes.data_as(C.POINTER(C.c_uint8))))
    return out


def build_bvh(prims, n, max_leaf=4, extra_point=None):
    nodes, order = C.POINTER(abi.LrBvhNode)(), C.POINTER(C.c_int32)()
    nn, depth = C.c_int(), C.c_int()
    ep = (C.c_float * 3)(*extra_point) if extra_point is not None else None
    _check(lib().lr_host_build_bvh(prims, n, max_leaf, ep, C.byref(nodes), C.byref(nn), C.byref(order), C.byref(depth)))
    try:
        node_arr = (abi.LrBvhNode * nn.value)()
        C.memmove(node_arr, nodes, C.sizeof(abi.LrBvhNode) * nn.value)
        order_arr = (C.c_int32 * max(n, 1))()
        if n:
            C.memmove(order_arr, order, 4 * n)
        return node_arr, nn.value, order_arr, depth.value
    finally:
        lib().lr_host_free(nodes)
        lib().lr_host_free(order)
