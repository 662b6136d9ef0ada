"""Fuzzing the host-side parsers (lumillyrender_amd/host/: toml_lite.h, scene_config.cpp, mesh_obj.cpp, image_io.cpp).

The reference panics on bad input (description.rs:34,38,139,156,178: unwrap / expect on the scene file, the TOML, every OBJ
and HDR it names).  The C++ host replaces that with error codes: whatever bytes arrive, every entry point returns LR_OK or an
LR_E* code with a message -- never a signal, never an exception other than LumillyError, never an allocation sized by a number
the input made up.  The mutations start from the scene files the loader is known to accept (this repo's scenes/*.toml and, where
/root/reference exists, the reference's nine) and from the Cornell OBJ / a small Radiance file, so most of them get past the
first token and die deep inside a parser.

The same file runs under AddressSanitizer + UBSan through tests/test_sanitizers.py (LR_HOST_LIB = the `make asan` build),
where an out-of-bounds read that happens not to crash still fails the run.  LR_FUZZ_EXAMPLES scales the number of inputs."""
import ctypes as C
import glob
import os
import shutil

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from lumillyrender_amd import abi, host
from tests.conftest import ROOT

N = int(os.environ.get("LR_FUZZ_EXAMPLES", "120"))
SETTINGS = dict(max_examples=N, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck), database=None)

SCENES = sorted(glob.glob(os.path.join(ROOT, "scenes", "*.toml"))) + sorted(glob.glob("/root/reference/scenes/*.toml"))
SCENE_TEXTS = [open(p, encoding="utf-8", errors="replace").read() for p in SCENES]
HOSTILE_NUMBERS = ["0", "-1", "-0", "1e39", "-1e39", "nan", "inf", "-inf", "1e-46", "2147483648", "-2147483649", "18446744073709551616",
                   "99999999999999999999999999", "0x10", "1_000", "1e", ".", "+", "--3", "007", "1.2.3"]
HOSTILE_TOKENS = ['"', "'", "[", "]", "[[", "]]", "{", "}", "=", ",", "\n", "\\", "#", "\x00", "\xff", "\t", '"""', "= =", "[[object]]", "[camera]",
                  'type = "obj"', 'type = ""', "transform = [", 'path = "/dev/zero"', 'path = ""', 'path = "../../../../etc/passwd"', "samples = -5",
                  "resolution = [0, 0]", "resolution = [100000, 100000]", "resolution = [-4, 8]", "fov = 0", "fov = 180", "radius = -1", "radius = 1e39"]


def load_ok_or_code(text, asset_root=host.ASSET_ROOT):
    """The loader's whole contract for arbitrary text: a Description, or LumillyError carrying an LR_E* code."""
    try:
        d = host.Description(text=text, asset_root=asset_root)
    except host.LumillyError as e:
        assert e.code in (abi.LR_EINVAL, abi.LR_ENOMEM, abi.LR_EUNSUPPORTED, abi.LR_EIO), e
        assert str(e)                                   # a message, not an empty string
        return None
    # what loads must be usable: the description's arrays are addressable up to their stated sizes
    desc = d.desc
    assert desc.n_prims >= 0 and desc.n_materials >= 0
    if desc.n_prims:
        _ = desc.prims[desc.n_prims - 1].v[8]
        assert 0 <= desc.prims[desc.n_prims - 1].material < desc.n_materials
    if desc.n_bvh_nodes:
        _ = desc.bvh_nodes[desc.n_bvh_nodes - 1].child[1]
    d.close()
    return True


@st.composite
def mutated_text(draw, texts):
    t = draw(st.sampled_from(texts))
    for _ in range(draw(st.integers(1, 4))):
        op = draw(st.integers(0, 8))
        if not t:
            break
        i = draw(st.integers(0, len(t)))
        j = min(len(t), i + draw(st.integers(0, 60)))
        if op == 0:                                          # truncation
            t = t[:i]
        elif op == 1:                                        # a span disappears
            t = t[:i] + t[j:]
        elif op == 2:                                        # a span is duplicated
            t = t[:j] + t[i:j] + t[j:]
        elif op == 3:                                        # a number becomes hostile
            import re
            nums = list(re.finditer(r"-?\d+\.?\d*(?:[eE][-+]?\d+)?", t))
            if nums:
                m = nums[draw(st.integers(0, len(nums) - 1))]
                t = t[:m.start()] + draw(st.sampled_from(HOSTILE_NUMBERS)) + t[m.end():]
        elif op == 4:                                        # a token lands somewhere
            t = t[:i] + draw(st.sampled_from(HOSTILE_TOKENS)) + t[i:]
        elif op == 5:                                        # a quote is lost (unterminated string)
            q = [k for k, c in enumerate(t) if c == '"']
            if q:
                k = q[draw(st.integers(0, len(q) - 1))]
                t = t[:k] + t[k + 1:]
        elif op == 6:                                        # random bytes
            t = t[:i] + draw(st.text(alphabet=st.characters(min_codepoint=0, max_codepoint=255), max_size=12)) + t[i:]
        elif op == 7:                                        # two lines swap
            ls = t.split("\n")
            if len(ls) > 2:
                a, b = draw(st.integers(0, len(ls) - 1)), draw(st.integers(0, len(ls) - 1))
                ls[a], ls[b] = ls[b], ls[a]
                t = "\n".join(ls)
        else:                                                # a line repeated many times (table arrays that grow)
            ls = t.split("\n")
            k = draw(st.integers(0, len(ls) - 1))
            t = "\n".join(ls[:k] + [ls[k]] * draw(st.integers(2, 40)) + ls[k:])
    return t


@settings(**SETTINGS)
@given(mutated_text(SCENE_TEXTS))
def test_mutated_scene_files_return_codes(text):
    load_ok_or_code(text)


def test_pathological_scene_texts():
    for text in ["", "\x00", "[", "]", "=", "a", "a=", "a=[", 'a="', "[[object]]\n" * 5000, "a = [" * 2000, "a = {" * 2000,
                 "a = " + "[" * 100000, "a = " + "9" * 100000, 'a = "' + "x" * 1000000 + '"', "\n" * 100000, "# only a comment",
                 "[renderer]\nsamples = 1\n[renderer]\nsamples = 2\n", "[film]\nresolution = [1, 1]\n" * 3]:
        load_ok_or_code(text)


# ---- OBJ / MTL ---------------------------------------------------------------------------------------------------------

OBJ_SCENE = '''mesh = [ { name = "m", type = "obj", path = "models/simple/quad.obj" } ]
material = [ { name = "w", type = "lambert", albedo = [0.5, 0.5, 0.5] } ]
[renderer]
integrator = "pt"
samples = 1
[film]
output = "png"
resolution = [8, 8]
[sky]
type = "uniform"
color = [1, 1, 1]
[camera]
type = "ideal-pinhole"
fov = 40
transform = [ { type = "look-at", origin = [0, 5, -5], target = [0, 0, 0], up = [0, 1, 0] } ]
[[object]]
mesh = "m"
material = "w"
'''
OBJ_BASES = [open(os.path.join(host.ASSET_ROOT, "models", "simple", n)).read() for n in ("quad.obj", "cbox.obj", "cbox_luminaire.obj")]
OBJ_LINES = ["f 1 2 3", "f -1 -2 -3", "f 0 0 0", "f 1 2", "f 1", "f", "f 1/1/1 2/2/2 3/3/3", "f 1//1 2//2 3//3", "f 1/ 2/ 3/", "f 999999999 1 2", "f -999999999 1 2",
             "f 1 2 3 4 5 6 7 8 9 10 11 12", "f 2147483648 1 2", "f a b c", "f 1.5 2.5 3.5", "v", "v 1", "v 1 2", "v nan nan nan", "v 1e39 0 0", "v x y z", "vn 0 1 0", "vt 0 0",
             "usemtl", "usemtl nosuchmaterial", "mtllib", "mtllib nosuchfile.mtl", "mtllib ../../../etc/passwd", "g", "o", "s off", "l 1 2", "p 1", "\x00", "f 1 2 3 " * 2000]


@pytest.fixture(scope="module")
def asset_copy(tmp_path_factory):
    root = tmp_path_factory.mktemp("assets")
    os.makedirs(root / "models" / "simple")
    for f in glob.glob(os.path.join(host.ASSET_ROOT, "models", "simple", "*")):
        shutil.copy(f, root / "models" / "simple")
    return root


@st.composite
def mutated_obj(draw):
    t = draw(st.sampled_from(OBJ_BASES))
    ls = t.split("\n")
    for _ in range(draw(st.integers(1, 5))):
        k = draw(st.integers(0, len(ls)))
        op = draw(st.integers(0, 3))
        if op == 0:
            ls.insert(k, draw(st.sampled_from(OBJ_LINES)))
        elif op == 1 and ls:
            del ls[min(k, len(ls) - 1)]
        elif op == 2 and ls:
            k = min(k, len(ls) - 1)
            ls[k] = ls[k][:draw(st.integers(0, max(0, len(ls[k]))))]
        else:
            ls = ls[:k]
    return "\n".join(ls)


@settings(**SETTINGS)
@given(mutated_obj(), mutated_obj())
def test_mutated_obj_and_mtl_files_return_codes(asset_copy, obj, mtl_like):
    (asset_copy / "models" / "simple" / "quad.obj").write_text(obj, encoding="utf-8", errors="replace")
    # the .mtl gets OBJ-ish garbage and real .mtl lines mixed: Kd with too few / hostile numbers, newmtl without a name
    (asset_copy / "models" / "simple" / "quad.mtl").write_text("newmtl quad\nKd 0.5 0.5\n" + mtl_like[:200] + "\nnewmtl\nKd nan 1e39 -1\n", encoding="utf-8", errors="replace")
    load_ok_or_code(OBJ_SCENE, asset_root=str(asset_copy))


# ---- Radiance .hdr -----------------------------------------------------------------------------------------------------

def _valid_hdr(tmp, w=16, h=8):
    img = np.random.default_rng(3).random((h, w, 3), dtype=np.float32) * 50
    path = os.path.join(tmp, "base.hdr")
    host.save_hdr(path, img)
    return open(path, "rb").read()


@pytest.fixture(scope="module")
def hdr_base(tmp_path_factory):
    return _valid_hdr(str(tmp_path_factory.mktemp("hdr")))


def load_hdr_ok_or_code(path):
    p, w, h = C.POINTER(C.c_float)(), C.c_int(), C.c_int()
    rc = host.lib().lr_host_load_hdr(os.fspath(path).encode(), C.byref(p), C.byref(w), C.byref(h))
    if rc < 0:
        assert rc in (abi.LR_EINVAL, abi.LR_ENOMEM, abi.LR_EUNSUPPORTED, abi.LR_EIO), rc
        assert host.lib().lr_host_last_error()
        return
    assert 0 < w.value <= 1 << 16 and 0 < h.value <= 1 << 16
    a = np.ctypeslib.as_array(p, shape=(h.value, w.value, 3))
    _ = float(a[-1, -1, 2])                                # the last texel is addressable
    host.lib().lr_host_free(p)


HDR_HEADERS = [b"-Y 8 +X 16\n", b"-Y 0 +X 16\n", b"-Y 8 +X 0\n", b"-Y -8 +X 16\n", b"-Y 1000000000 +X 1000000000\n", b"-Y 99999999999999999999 +X 16\n", b"+X 16 -Y 8\n",
               b"-Y 8\n", b"-Y 8 +X\n", b"-Y 8 +X 16", b"\n", b"-Y 65536 +X 131072\n", b"-Y 8 +X 32768\n"]


@settings(**SETTINGS)
@given(st.data())
def test_mutated_hdr_files_return_codes(tmp_path_factory, hdr_base, data):
    b = bytearray(hdr_base)
    for _ in range(data.draw(st.integers(1, 4))):
        op = data.draw(st.integers(0, 4))
        i = data.draw(st.integers(0, len(b)))
        if op == 0:
            b = b[:i]
        elif op == 1 and len(b):
            k = min(i, len(b) - 1)
            b[k] = data.draw(st.integers(0, 255))
        elif op == 2:
            b = b[:i] + bytearray(data.draw(st.lists(st.integers(0, 255), max_size=16))) + b[i:]
        elif op == 3:                                          # the resolution line is replaced
            k = b.find(b"-Y")
            if k >= 0:
                e = b.find(b"\n", k)
                b = b[:k] + bytearray(data.draw(st.sampled_from(HDR_HEADERS))) + b[e + 1:]
        else:                                                  # the magic line
            e = b.find(b"\n")
            b = bytearray(data.draw(st.sampled_from([b"#?RADIANCE", b"#?RGBE", b"", b"#?", b"RADIANCE", b"\x00\x00"]))) + b[e:]
    path = tmp_path_factory.getbasetemp() / "fuzz.hdr"
    path.write_bytes(bytes(b))
    load_hdr_ok_or_code(path)


def test_missing_and_special_files_return_codes(tmp_path):
    for p in [tmp_path / "nope.hdr", tmp_path, "/dev/null", "/proc/self/mem"]:
        load_hdr_ok_or_code(p)
    for p in [tmp_path / "nope.toml", tmp_path, "/dev/null"]:
        with pytest.raises(host.LumillyError):
            host.Description(os.fspath(p))
