"""The host library and the oracle under AddressSanitizer + UndefinedBehaviorSanitizer (CPU builds only: GPU sanitizers are not
available on this pool).  `make -C lumillyrender_amd/host asan` and `make -C oracle asan` build the same sources with
-fsanitize=address,undefined; a child pytest with libasan preloaded loads them through LR_HOST_LIB / LR_ORACLE_LIB and runs the
loader suite, the parser fuzz (more inputs than the plain run), the oracle's known-answer / property / golden suites and the
multi-rank tile assembly.  Any report -- heap overflow, use after free, misaligned or out-of-range access -- aborts the child
and fails this test.  (Leak checking is off: CPython itself does not free everything at exit.)"""
import os
import subprocess
import sys

import pytest

from tests.conftest import ROOT


def _asan_runtime():
    """libasan AND libstdc++, in that order: the interpreter links neither, and AddressSanitizer resolves its __cxa_throw
    interceptor when it starts -- without libstdc++ in the process by then, the first C++ exception inside the library
    (every LR_E* return starts as one) aborts."""
    libs = []
    for name in ("libasan.so", "libstdc++.so.6"):
        out = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
        if not (out and os.path.isabs(out) and os.path.exists(out)):
            return None
        libs.append(os.path.realpath(out))
    return " ".join(libs)


def _run_under_asan(test_args, extra_env=None, timeout=1500):
    rt = _asan_runtime()
    if rt is None:
        pytest.skip("no libasan next to gcc")
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "lumillyrender_amd", "host"), "asan"], check=True)
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"], check=True)
    env = dict(os.environ,
               LD_PRELOAD=rt,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1:allocator_may_return_null=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               LR_HOST_LIB=os.path.join(ROOT, "lumillyrender_amd", "liblumilly_host_asan.so"),
               LR_ORACLE_LIB=os.path.join(ROOT, "oracle", "liboracle_asan.so"),
               PYTHONMALLOC="malloc")
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "-m", "not gpu"] + test_args,
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    tail = (r.stdout + r.stderr)[-6000:]
    assert "AddressSanitizer" not in r.stdout + r.stderr and "runtime error:" not in r.stdout + r.stderr, tail
    assert r.returncode == 0, tail
    return r.stdout


def test_host_loader_and_parser_fuzz_under_asan():
    out = _run_under_asan(["tests/test_host_loader.py", "tests/test_host_fuzz.py"], {"LR_FUZZ_EXAMPLES": "300"})
    assert " passed" in out and "failed" not in out


def test_oracle_suites_under_asan():
    # (LUMILLY_TEST_LIGHT: the two suites cut their largest inputs -- 4 * 10^5 edge rays, the stated-size tiles at 1024 / 4096 spp -- by 20x;
    #  the instrumented oracle is 15x slower and every code path is reached by the smaller inputs)
    out = _run_under_asan(["tests/test_oracle_kat.py", "tests/test_oracle_properties.py", "tests/test_golden_fixtures.py"], {"LUMILLY_TEST_LIGHT": "1"})
    assert " passed" in out and "failed" not in out


def test_multirank_tile_assembly_under_asan():
    """tests/test_multirank_gloo.py spawns torch.distributed ranks; torch itself is not instrumented, the host library the ranks
    cut tiles with is."""
    out = _run_under_asan(["tests/test_multirank_gloo.py", "-k", "shared_memory_film or killed or subgroup"], timeout=1800)
    assert " passed" in out and "failed" not in out
