"""lumillyrender_amd -- MI355X-native replacement for LumillyRender's per-pixel sampling loop.

Layout (only what the hot path needs):
  csrc/   hand-written HIP kernels for gfx950 + the C ABI (include/lumilly_hip.h) -> liblumilly_hip.so
  host/   C++ host: scene loader, SAH BVH build, tile queue, png/hdr (include/lumilly_host.h) -> liblumilly_host.so
  host.py / device.py  thin ctypes bindings used by tests, bench.py and the multi-GPU driver
"""
from . import abi  # noqa: F401

__all__ = ["abi"]
