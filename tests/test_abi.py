"""The C-ABI libraries load on a machine without a GPU and export exactly what include/*.h declares;
the ctypes struct mirrors have the C sizes.  (No compute calls here.)"""
import ctypes as C
import os
import re

from lumillyrender_amd import abi, host
from tests.conftest import ROOT


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lr_[a-z0-9_]+)\s*\(", text)))


def test_host_library_exports_header_symbols():
    lib = host.lib()
    names = declared_functions("lumilly_host.h")
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), n


def test_hip_library_exports_header_symbols():
    path = os.path.join(ROOT, "lumillyrender_amd", "liblumilly_hip.so")
    assert os.path.exists(path), "liblumilly_hip.so missing: run __graft_entry__.build()"
    lib = C.CDLL(path)                      # loads without a GPU; nothing is called that touches one
    names = declared_functions("lumilly_hip.h")
    assert {"lr_scene_create", "lr_render", "lr_render_device", "lr_scene_destroy", "lr_get_stats", "lr_last_error", "lr_device_count"} <= set(names)
    for n in names:
        assert hasattr(lib, n), n
    diag = declared_functions("lumilly_hip_diag.h")          # diagnostics live in their own header, outside the drop-in surface
    assert {"lr_selftest_math", "lr_selftest_intersect", "lr_selftest_brute", "lr_selftest_rng", "lr_selftest_rcp"} <= set(diag)
    assert not (set(diag) & set(names)), "a diagnostic entry point leaked into the product header"
    for n in diag:
        assert hasattr(lib, n), n
    lib.lr_build_info.restype = C.c_char_p
    assert b"gfx950" in lib.lr_build_info()


def test_struct_sizes_match_c():
    for name in ("LrCamera", "LrMaterial", "LrPrimitive", "LrSky", "LrBvhNode", "LrSceneDesc", "LrRenderParams", "LrTile",
                 "LrStats", "LrRendererConfig", "LrFilmConfig"):
        assert host.lib().lr_host_sizeof(name.encode()) == C.sizeof(getattr(abi, name)), name
    assert C.sizeof(abi.LrBvhNode) == 64 and C.sizeof(abi.LrPrimitive) == 48


def test_product_never_imports_the_oracle():
    """The render path must not reach into oracle/ (a product path routed through the checker would
    void every parity claim)."""
    pkg = os.path.join(ROOT, "lumillyrender_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "/oracle/" not in text and "oracle/lr_" not in text, (dirpath, f)
                assert "liboracle" not in text and "from oracle" not in text and "import oracle" not in text, (dirpath, f)
