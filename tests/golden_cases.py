"""What tests/golden/ holds and how it is produced -- shared by the generator (tests/golden/make_golden.py, runs the CPU oracle),
the CPU suite (the oracle must still reproduce every fixture: tests/test_golden_fixtures.py) and the GPU suite (the HIP path must
reproduce them WITHOUT the oracle: tests/test_gpu_film.py, tests/test_gpu_math.py).

Nothing can be captured from the reference itself (Rust, no toolchain, unseeded RNG: SURVEY 8c), so the vectors are the oracle's
own output at fixed seeds.  A fixture survives an oracle edit that would move both sides of a live comparison together.

  film crops   32x24-class crops of every scene class at 8 spp (primitive / new-cbox pt + pt-direct / brdf with GGX,
               Phong, Blinn-Phong / the 100k-triangle mesh / thin lens + IBL)
  functions    per-function vectors for SURVEY 8(a) rows a9-a19 and the cameras: inputs + outputs in one .npz
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
FUNCTIONS = os.path.join(GOLDEN, "functions.npz")


def _phong_edit(mat):
    """scenes/brdf-row.toml with the GGX row replaced by Phong / Blinn-Phong lobes, alpha in {1, 5, 10, 20}."""
    def edit(t):
        out, k = [], 0
        alphas = ["1", "5", "10", "20"]
        for ln in t.split("\n"):
            if ln.strip() == 'type = "ggx"':
                out.append(f'type = "{mat}"')
            elif ln.startswith("roughness"):
                out.append(f"alpha = {alphas[k]}"); k += 1
            elif ln.startswith("ior"):
                pass
            else:
                out.append(ln)
        return "\n".join(out)
    return edit


def _omni_edit(t):
    return t.replace('type = "ideal-pinhole"\nfov = 39.3077\n', 'type = "omnidirectional"\n').replace("[278, 273, -800]", "[278, 273, 100]")


EDITS = {None: None, "phong": _phong_edit("phong"), "blinn-phong": _phong_edit("blinn-phong"), "omni": _omni_edit}

# (scene, edit, W, H, spp, integrator or None = the scene's, seed, needs generated assets)
FILM_CASES = [
    ("two-spheres.toml", None, 16, 16, 8, None, 3, False),
    ("cbox-spheres.toml", None, 16, 16, 8, 0, 3, False),
    ("cbox-spheres.toml", None, 16, 16, 8, 1, 3, False),
    ("brdf-row.toml", None, 32, 18, 8, 1, 3, False),
    ("brdf-row.toml", "phong", 32, 18, 8, 1, 3, False),
    ("brdf-row.toml", "blinn-phong", 32, 18, 8, 1, 3, False),
    ("mesh-box.toml", None, 32, 24, 8, 0, 3, True),
    ("ibl-lens.toml", None, 32, 24, 8, 1, 3, True),
]


# The four GPU configs of BASELINE.json AT THEIR STATED spp on an 8 x 6 film (round 5): main.rs:92-121 folds spp samples per pixel
# in one pass; the device cuts them into chunks whose schedule changes with spp (lumilly_hip.hip chunk_schedule: body + taper,
# 16-sample chunks up to 1024 spp, 32-sample chunks beyond) and k_resolve adds up to ~300 chunk sums per pixel -- the branch every
# stated config takes and no 8-spp crop reaches.  Generated with the oracle's reference-literal tree (bvh.rs:131-141, mode BVH,
# pad 0; the brute-force definition gives the same bits on these films and takes minutes on the 100k-triangle scenes).
STATED_SPP_CASES = [
    ("cbox-spheres.toml", None, 8, 6, 1024, 1, 5, False),       # configs[1]
    ("brdf-row.toml", None, 8, 6, 4096, 1, 5, False),           # configs[2]
    ("mesh-box.toml", None, 8, 6, 2048, 0, 5, True),            # configs[3]
    ("ibl-lens.toml", None, 8, 6, 8192, 1, 5, True),            # configs[4]
]


# BASELINE.json configs[0..4] AT THEIR STATED FILM SIZE and spp (round 6): main.rs:80-122 runs per pixel of the W x H film, and
# pixel footprints, the thin lens' sensor_pixel_area (camera.rs:391) and which geometric edges a pixel straddles all depend on
# the resolution -- an 8 x 6 film of the same scene sees none of it (the own-box semantics of bvh.rs:20-25 showed at 1024 x 1024
# only).  RNG keys are (pixel index, sample), so the oracle can render SCATTERED TILES of the full-size film: 64 tiles of 16 x 16
# pixels, one per cell of an 8 x 8 grid at a position that differs from cell to cell (16 384 pixels, 196 KB as .npy), plus -- for
# configs[1] -- the whole pixel rows through the images of the box's edges (rows 114, 231, 419, 1001: where round 5's closest
# hit over ALL primitives left the reference).  name -> (scene, edit, W, H, spp, integrator, seed, needs assets, extra whole rows)
STATED_SIZE_CASES = {
    "c1": ("cbox-spheres.toml", None, 256, 256, 16, 0, 0, False, ()),
    "c2": ("cbox-spheres.toml", None, 1024, 1024, 1024, 1, 0, False, (114, 231, 419, 1001)),
    "c3": ("brdf-row.toml", None, 960, 540, 4096, 1, 0, False, ()),
    "c3p": ("brdf-row.toml", "phong", 960, 540, 4096, 1, 0, False, ()),
    "c3b": ("brdf-row.toml", "blinn-phong", 960, 540, 4096, 1, 0, False, ()),
    "c4": ("mesh-box.toml", None, 1920, 1370, 2048, 0, 0, True, ()),
    "c5": ("ibl-lens.toml", None, 2048, 2048, 8192, 1, 0, True, ()),
}


def stated_tiles(w, h, rows=()):
    """[(x0, y0, tw, th)]: disjoint; 8 x 8 scattered 16-px tiles + whole rows (a tile that a row crosses is dropped)."""
    out = []
    cw, ch = w // 8, h // 8
    for j in range(8):
        for i in range(8):
            x0 = cw * i + (37 * j + 11 * i) % max(1, cw - 16)
            y0 = ch * j + (29 * i + 17 * j) % max(1, ch - 16)
            if any(y0 <= r < y0 + 16 for r in rows):
                continue
            out.append((x0, y0, 16, 16))
    out += [(0, r, w, 1) for r in rows]
    return out


def tile_array(tl):
    from lumillyrender_amd import abi
    tiles = (abi.LrTile * len(tl))()
    for k, (x, y, tw, th) in enumerate(tl):
        tiles[k].x0, tiles[k].y0, tiles[k].w, tiles[k].h = x, y, tw, th
    return tiles


def tile_mask(w, h, tl):
    m = np.zeros((h, w), dtype=bool)
    for x, y, tw, th in tl:
        assert not m[y:y + th, x:x + tw].any(), "tiles overlap"
        m[y:y + th, x:x + tw] = True
    return m


def pack_tiles(img, tl):
    """The tiles' pixels in tile order, rows of a tile top to bottom: (n_pix, 3)."""
    return np.concatenate([img[y:y + th, x:x + tw].reshape(-1, 3) for x, y, tw, th in tl], axis=0)


def stated_name(key):
    return f"stated_{key}.npy"


# phong.rs:47-68 at alpha = 20: the six pixels of the stated Phong row (960 x 540 x 4096 spp, seed 0) in which one sample draws a direction
# whose pdf 22 / 2 pi * c^20 is a DENORMAL (not 0): (x, y, index of that sample).  scene.rs:101 divides the equally small BRDF value by it and
# stays finite; found by the whole-frame check of round 6 (a hardware reciprocal reads a denormal as 0).  The fixture holds the oracle's
# pixel after samples 0 .. k (RNG keys are (pixel, sample): a prefix of the stated samples).
DENORMAL_PDF_PIXELS = [(580, 182, 1081), (510, 268, 3170), (291, 291, 1513), (544, 358, 3795), (432, 424, 3584), (476, 477, 3223)]
DENORMAL_PDF_FIXTURE = "denormal_pdf_c3p.npy"


def one_pixel_tile(x, y):
    from lumillyrender_amd import abi
    t = (abi.LrTile * 1)()
    t[0].x0, t[0].y0, t[0].w, t[0].h = int(x), int(y), 1, 1
    return t


def film_name(case):
    name, edit, w, h, spp, integ, seed, _ = case
    tag = "" if edit is None else "_" + edit
    return f"{name.replace('.toml', '')}{tag}_{w}x{h}_{spp}spp_i{integ}_s{seed}.npy"


def have_generated_assets():
    from lumillyrender_amd import host
    return os.path.exists(os.path.join(host.ASSET_ROOT, "models/blob/blob.obj")) and os.path.exists(os.path.join(host.ASSET_ROOT, "models/ibl/sky_3k.hdr"))


def load_scene(name, edit, w, h):
    from lumillyrender_amd import host
    path = os.path.join(ROOT, "scenes", name)
    d = host.Description(path) if EDITS[edit] is None else host.Description(text=EDITS[edit](open(path).read()))
    d.set_resolution(w, h)
    return d


# ---- function vectors: the INPUTS (fixed seeds; regenerated identically by every consumer that wants to double-check them) ----

MATERIALS = {        # name: (type, color, param0, param1)   -- lumilly_hip.h LR_MAT_*
    "lambert": (0, (0.74, 0.74, 0.73), 0.0, 0.0),
    "phong": (1, (0.9, 0.8, 0.7), 10.0, 0.0),
    "blinn_phong": (2, (0.9, 0.8, 0.7), 20.0, 0.0),
    "ggx": (3, (0.95, 0.93, 0.88), 0.4, 1.0e5),
    "ideal_refraction": (4, (0.9, 0.95, 0.99), 1.5, 0.01),
}


def material(name):
    from lumillyrender_amd import abi
    t, color, p0, p1 = MATERIALS[name]
    m = abi.LrMaterial()
    m.type = t
    m.color[:] = color
    m.param[0], m.param[1] = p0, p1
    return m


def _unit(v):
    return (v / np.linalg.norm(v, axis=-1, keepdims=True)).astype(np.float32)


def material_inputs(n=64, seed=11):
    """(n, 13): out_ (unit, on the normal's side for most rows), normal (unit), position, xi[3], fly distance."""
    rng = np.random.default_rng(seed)
    nrm = _unit(rng.standard_normal((n, 3)))
    out_ = _unit(rng.standard_normal((n, 3)))
    flip = np.sum(out_ * nrm, axis=1) < 0
    keep_below = np.arange(n) % 8 == 7                                   # every eighth row looks at the back face (orienting_normal)
    out_[flip & ~keep_below] *= -1
    pos = (rng.uniform(-500, 500, (n, 3))).astype(np.float32)
    xi = rng.random((n, 3), dtype=np.float32)
    dist = rng.uniform(0.5, 300.0, (n, 1)).astype(np.float32)
    return np.concatenate([out_, nrm, pos, xi, dist], axis=1).astype(np.float32)


def aabb_inputs(n=64, seed=12):
    rng = np.random.default_rng(seed)
    lo = rng.uniform(-100, 100, (n, 3)); ext = rng.uniform(1, 80, (n, 3))
    box = np.concatenate([lo, lo + ext], axis=1).astype(np.float32)
    o = rng.uniform(-300, 300, (n, 3)).astype(np.float32)
    aim = lo + ext * rng.uniform(-0.6, 1.6, (n, 3))                     # about half of the rays hit
    d = _unit(aim - o)
    d[::9, 0] = 0.0                                                       # axis-parallel rays: the divisions by zero of aabb.rs:77
    return box, o, d


def rays_in_box(n, lo, hi, seed):
    rng = np.random.default_rng(seed)
    o = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    return o, _unit(rng.standard_normal((n, 3)))


def rays_at(n, centre, radius, spread, seed):
    """origins on a sphere around `centre`, aimed at points within `spread` of it (most of them hit what sits there)."""
    rng = np.random.default_rng(seed)
    c = np.asarray(centre, dtype=np.float64)
    o = np.clip(c + _unit(rng.standard_normal((n, 3))).astype(np.float64) * radius, 5.0, 545.0)    # inside the Cornell box (the envelope the padded boxes are built for)
    aim = c + rng.uniform(-1, 1, (n, 3)) * np.asarray(spread, dtype=np.float64)
    return o.astype(np.float32), _unit(aim - o)


MESH_RAYS = dict(n=512, centre=(240.0, 95.0, 231.0), radius=170.0, spread=(55.0, 90.0, 50.0), seed=22)    # the blob of scenes/mesh-box.toml


def sky_directions(seed=13, n=256):
    rng = np.random.default_rng(seed)
    axes = np.array([[0, 1, 0], [0, -1, 0], [1, 0, 0], [-1, 0, 0], [0, 0, 1], [0, 0, -1]], dtype=np.float64)
    near = (axes[:, None, :] + rng.standard_normal((6, 16, 3)) * 1e-4).reshape(-1, 3)
    seam = np.stack([-np.ones(32), rng.uniform(-1, 1, 32), rng.standard_normal(32) * 1e-6], axis=1)
    return _unit(np.concatenate([rng.standard_normal((n, 3)), axes, near, seam]))


def camera_inputs(w, h, n=64, seed=14):
    rng = np.random.default_rng(seed)
    xy = np.stack([rng.integers(0, w, n), rng.integers(0, h, n)], axis=1).astype(np.int32)
    xy[:4] = [[0, 0], [w - 1, 0], [0, h - 1], [w - 1, h - 1]]
    return xy, rng.random((n, 4), dtype=np.float32)


CAMERA_SCENES = {"pinhole": ("cbox-spheres.toml", None, False), "omnidirectional": ("cbox-spheres.toml", "omni", False), "thin_lens": ("ibl-lens.toml", None, True)}

MATH_CASES = {       # fn name -> (selftest id, unary?)
    "sin": (0, True), "cos": (1, True), "acos": (2, True), "atan2": (3, False),
}


def math_inputs(name, seed=15):
    rng = np.random.default_rng(seed + MATH_CASES[name][0])
    if name in ("sin", "cos"):
        a = np.concatenate([rng.uniform(0, 2 * np.pi, 48), np.arange(9) * (np.pi / 4), [1e-8, 6.2831855]]).astype(np.float32)
        return a, None
    if name == "acos":
        a = np.concatenate([rng.uniform(-1, 1, 48), [-1, 1, 0, -0.5, 0.5, 0.50000006, -0.50000006, 1e-5, -1e-5, 0.99999994]]).astype(np.float32)
        return a, None
    y = np.concatenate([rng.standard_normal(48), [0, 0, 1, -1, 0.0, -0.0, 1e-30, -1e-30]]).astype(np.float32)
    x = np.concatenate([rng.standard_normal(48), [1, -1, 0, 0, 0.0, -1.0, -1, -1]]).astype(np.float32)
    return y, x
