"""bench.py attaches a rocprofv3 figure (HBM traffic, VALU instructions) to its JSON line only when the committed profile was
taken on THIS workload (a stale file must never be mixed with the current timing), and takes the newest one."""
import importlib.util
import json
import os

import pytest

from tests.conftest import ROOT


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_profile_is_attached_only_for_the_same_workload(tmp_path, monkeypatch):
    b = _bench()
    (tmp_path / "profiles").mkdir()
    monkeypatch.setattr(b, "ROOT", str(tmp_path))
    wl = {"config": "c2", "scene": "cbox-spheres.toml", "width": 1024, "height": 1024, "spp": 1024, "path_slots": 393216}
    json.dump({"workload": dict(wl, when=100), "k_resident_hbm_bytes_per_launch": 1}, open(tmp_path / "profiles" / "r02b_traffic_c2.json", "w"))
    json.dump({"workload": dict(wl, when=200), "k_resident_hbm_bytes_per_launch": 2}, open(tmp_path / "profiles" / "r02_traffic_c2.json", "w"))
    json.dump({"workload": dict(wl, spp=256, when=300), "k_resident_hbm_bytes_per_launch": 3}, open(tmp_path / "profiles" / "r03_traffic_c2.json", "w"))
    json.dump({"note": "no workload recorded", "k_resident_hbm_bytes_per_launch": 4}, open(tmp_path / "profiles" / "r04_traffic_c2.json", "w"))
    want = {"scene": "cbox-spheres.toml", "width": 1024, "height": 1024, "spp": 1024}
    path, d = b.load_profile("traffic", "c2", want)
    assert os.path.basename(path) == "r02_traffic_c2.json" and d["k_resident_hbm_bytes_per_launch"] == 2      # newest stamp wins, not the last name
    assert b.load_profile("traffic", "c2", dict(want, spp=512)) is None                                          # another spp: nothing attached
    assert b.load_profile("traffic", "c2", dict(want, scene="brdf-row.toml")) is None
    assert b.load_profile("traffic", "c3", want) is None
    assert b.load_profile("pmc", "c2", want) is None


def test_committed_profiles_name_their_workload():
    """Every traffic / pmc file the bench may pick up records scene, film size and spp or slot count."""
    import glob
    files = glob.glob(os.path.join(ROOT, "profiles", "r*_traffic_c*.json")) + glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_c*.json"))
    assert files
    for f in files:
        wl = json.load(open(f)).get("workload", {})
        assert {"scene", "width", "height", "spp"} <= set(wl), f
    b = _bench()
    for cfg, (scene, w, h, spp, *_rest) in b.CONFIGS.items():
        if cfg in b.CPU_ONLY_CONFIGS:
            continue
        want = {"scene": scene, "width": w, "height": h}
        assert b.load_profile("traffic", cfg, want) is not None, cfg          # each BASELINE config has its HBM report


def _tree_build_id():
    import subprocess, sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import build_id
    flags = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "lumillyrender_amd", "csrc"), "print-flags"], capture_output=True, text=True, check=True).stdout.strip()
    return build_id.build_id(flags)


def test_library_build_id_is_the_content_hash_of_the_sources():
    """lr_build_info() carries sha256(csrc/* + the C-ABI headers)[:16] + '-' + sha256(flags)[:8] (tools/build_id.py, written by the
    Makefile): the in-tree library is the one the sources in the tree build, and a profile that records this id names its library."""
    from lumillyrender_amd import device
    info = device.build_info()
    assert "gfx950" in info and "build=" in info
    assert device.build_id() == _tree_build_id(), "liblumilly_hip.so is stale against lumillyrender_amd/csrc (run make -C lumillyrender_amd/csrc)"


def test_newest_profiles_were_taken_on_this_build():
    """`roofline` mixes a live launch duration with the instruction counts of a COMMITTED counter pass, so the
    pass must have been taken on the library that is being timed.  Every profiles/*_pmc_* / *_traffic_* file records the build id
    of the library that was profiled (tools/profile_round.sh); the newest one of every BASELINE config must carry the id of the
    sources in this tree -- a kernel edit without a new profile pass fails here, before a stale count reaches a bench line."""
    b = _bench()
    want_build = _tree_build_id()
    for cfg, (scene, w, h, spp, *_rest) in b.CONFIGS.items():
        if cfg in b.CPU_ONLY_CONFIGS:
            continue
        for kind in ("pmc", "traffic"):
            got = b.load_profile(kind, cfg, {"scene": scene, "width": w, "height": h, "spp": spp})
            assert got is not None, (cfg, kind)
            assert got[1].get("workload", {}).get("build") == want_build, (cfg, kind, os.path.basename(got[0]), got[1].get("workload", {}).get("build"), want_build)


def test_bench_refuses_product_changing_environment(tmp_path):
    """A stale LR_* variable in the shell must not silently change what bench.py measures.  Since round 6 the
    PRODUCT library reads no LR_* switch at all (csrc/lr_knobs.h: the knobs are compiled into the knob build only), so what is left to
    refuse are the variables that select another library build; every LR_* variable is recorded."""
    import subprocess, sys
    env = dict(os.environ, LR_HIP_LIB=os.path.join(ROOT, "lumillyrender_amd", "liblumilly_hip_knobs.so"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1"], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode != 0 and "LR_HIP_LIB" in (r.stderr + r.stdout) and "--allow-overrides" in (r.stderr + r.stdout)
    b = _bench()
    assert set(b.PRODUCT_ENV) == {"LR_HIP_LIB", "LR_HOST_LIB", "LR_ORACLE_LIB"}
    # the library sources: every LR_* variable is read through lr_knob() (compiled to "unset" in the product build) or is LR_DEBUG (prints only)
    import re
    src = ""
    for f in ("lumillyrender_amd/csrc/lumilly_hip.hip", "lumillyrender_amd/csrc/lr_lbvh.hip", "lumillyrender_amd/csrc/lr_kernels.h", "lumillyrender_amd/csrc/lr_path.h"):
        src += open(os.path.join(ROOT, f)).read()
    assert set(re.findall(r'getenv\("(LR_[A-Z0-9_]+)"\)', src)) <= {"LR_DEBUG"}
    assert len(set(re.findall(r'lr_knob\("(LR_[A-Z0-9_]+)"\)', src))) >= 10
    # ... and the python side reads library paths only
    py = open(os.path.join(ROOT, "lumillyrender_amd/device.py")).read() + open(os.path.join(ROOT, "lumillyrender_amd/host.py")).read() + open(os.path.join(ROOT, "oracle/binding.py")).read()
    assert set(re.findall(r'environ\.get\("(LR_[A-Z0-9_]+)"\)', py)) <= set(b.PRODUCT_ENV)


def test_product_library_has_no_knob_strings():
    """`strings liblumilly_hip.so | grep '^LR_'` is LR_DEBUG and nothing else; the knob build carries them."""
    import re

    def names(path):
        return set(m.decode() for m in re.findall(rb"(?<![A-Za-z0-9_])(LR_[A-Z][A-Z0-9_]+)\x00", open(path, "rb").read()))
    lib = os.path.join(ROOT, "lumillyrender_amd", "liblumilly_hip.so")
    knob = os.path.join(ROOT, "lumillyrender_amd", "liblumilly_hip_knobs.so")
    if not (os.path.exists(lib) and os.path.exists(knob)):
        pytest.skip("libraries not built")
    assert names(lib) <= {"LR_DEBUG"}, names(lib)
    assert {"LR_PIPELINE", "LR_BAND_PIX", "LR_STACK_LDS", "LR_SUB_SHIFT"} <= names(knob)
