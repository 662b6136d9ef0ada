import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # CPU-side natives (host library, oracle) are cheap to (re)build; the HIP library is built by
    # __graft_entry__.build() and travels to the GPU box as a prebuilt .so
    for sub in ("lumillyrender_amd/host", "oracle"):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, sub)], check=True)


@pytest.fixture(scope="session")
def root():
    return ROOT


def scene_path(name):
    return os.path.join(ROOT, "scenes", name)


@pytest.fixture(scope="module")
def dev():
    """The device binding (GPU suites).  There is no CPU fallback: without a GPU this fails."""
    from lumillyrender_amd import device
    assert device.device_count() >= 1, "no HIP device: the product path has no CPU fallback"
    return device


@pytest.fixture(scope="module")
def oracle():
    from oracle import binding
    return binding


@pytest.fixture
def knobs(monkeypatch):
    """The tests that steer the library with an LR_* variable (stack placement, banding, pipeline variants, builders) run on the KNOB
    build: the product library compiles those variables out (csrc/lr_knobs.h).  Swaps the library behind lumillyrender_amd.device for
    the duration of the test; scenes must be created and closed inside it."""
    from lumillyrender_amd import device
    klib = device.load_library(device.KNOBS_LIB_PATH)
    assert b"knobs=on" in klib.lr_build_info()
    assert "knobs=off" in device.build_info()
    monkeypatch.setattr(device, "_lib", klib)
    return monkeypatch
