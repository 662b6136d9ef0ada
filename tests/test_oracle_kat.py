"""The oracle against every known-answer test the reference holds for the hot path.

Each test restates one #[test] of the reference (file:line in the docstring) with the same inputs
and the same assertion; these are the only reference-side vectors that exist for this path.
"""
import ctypes as C
import math

import numpy as np

from lumillyrender_amd import abi
from oracle import binding as oracle

EPS = 1e-3
PI = np.float32(3.14159265358979323846)
f3 = oracle.f3


def norm(v):
    return float(np.linalg.norm(np.asarray(v, dtype=np.float64)))


def normalize(v):
    v = np.asarray(v, dtype=np.float32)
    return (v / np.float32(math.sqrt(float(np.dot(v, v))))).astype(np.float32)


TRI = [5.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 5.0]


def tri(o, d, variant):
    out = (C.c_float * 7)()
    hit = oracle.lib().lr_oracle_triangle_intersect(f3(TRI), f3(o), f3(d), variant, out)
    return (hit, np.array(out[:], dtype=np.float32))


def test_triangle_intersect_mt_front():
    """triangle.rs:157-175"""
    h1, i1 = tri([1, 5, 1], [0, -1, 0], 1)
    h2, i2 = tri([1, 5, 1], [0, -1, 0], 0)
    assert h1 and h2
    assert norm(i1[4:7] - i2[4:7]) < 1e-3 and norm(i1[1:4] - i2[1:4]) < 1e-3 and abs(i1[0] - i2[0]) < 1e-3
    assert abs(i2[0] - 5.0) < 1e-6 and norm(i2[1:4] - np.array([1, 0, 1])) < 1e-6


def test_triangle_intersect_mt_back():
    """triangle.rs:177-195"""
    h1, i1 = tri([1, -5, 1], [0, 1, 0], 1)
    h2, i2 = tri([1, -5, 1], [0, 1, 0], 0)
    assert h1 and h2
    assert norm(i1[4:7] - i2[4:7]) < 1e-3 and norm(i1[1:4] - i2[1:4]) < 1e-3 and abs(i1[0] - i2[0]) < 1e-3


def test_triangle_intersect_3c_near():
    """triangle.rs:197-215: a ray leaving the surface does not re-hit it (t < EPS)."""
    h1, i1 = tri([1, 5, 1], [0, -1, 0], 1)
    assert h1
    h2, _ = tri(i1[1:4], [0, 1, 0], 1)
    assert not h2


def test_triangle_intersect_mt_near():
    """triangle.rs:217-235"""
    h1, i1 = tri([1, 5, 1], [0, -1, 0], 0)
    assert h1
    h2, _ = tri(i1[1:4], [0, 1, 0], 0)
    assert not h2


def reflect(v, n):
    out = (C.c_float * 3)()
    oracle.lib().lr_oracle_reflect(f3(v), f3(n), out)
    return np.array(out[:], dtype=np.float32)


def refract(v, n, ratio):
    out = (C.c_float * 3)()
    ok = oracle.lib().lr_oracle_refract(f3(v), f3(n), ratio, out)
    return np.array(out[:], dtype=np.float32) if ok else None


def test_reflect():
    """util.rs:49-55"""
    v = normalize([1, 0, 1])
    r = reflect(v, [0, 0, 1])
    assert norm(r - normalize([-1, 0, 1])) < EPS


def test_refract_total_reflection():
    """util.rs:57-63"""
    v = normalize([1, 0, 0.1])
    assert refract(v, [0, 0, 1], 1.5 / 1.0) is None


def test_refract_snell():
    """util.rs:65-81: sin(t1)/sin(t2) == n1/n2 at 30 degrees, result of unit length."""
    n1, n2 = 1.0, 1.5
    t1 = 30.0 / 180.0 * math.pi
    v = normalize([math.tan(t1), 0, 1])
    r = refract(v, [0, 0, 1], 1.5 / 1.0)
    assert r is not None
    sin_t2 = norm(np.cross(r.astype(np.float64), [0, 0, -1]))
    # the reference test passes from_per_to_ior = 1.5 while naming n1 = 1, n2 = 1.5; keep its assertion literally
    assert abs(math.sin(t1) / sin_t2 - n1 / n2) < EPS
    assert abs(norm(r) - 1.0) < EPS


def refr_material(ior, absorb=0.0):
    m = abi.LrMaterial()
    m.type = abi.LR_MAT_IDEAL_REFRACTION
    m.color[:] = [1.0, 1.0, 1.0]
    m.param[0], m.param[1] = ior, absorb
    return m


def ior_pair(m, out_, n):
    p = (C.c_float * 2)()
    oracle.lib().lr_oracle_ior_pair(C.byref(m), f3(out_), f3(n), p)
    return float(p[0]), float(p[1])


def test_ior_pair_into():
    """ideal_refraction.rs:167-180"""
    m = refr_material(1.5)
    a, b = ior_pair(m, normalize([1, 0, 1]), [0, 0, 1])
    assert a == 1.0 and b == 1.5


def test_ior_pair_outgoing():
    """ideal_refraction.rs:182-195"""
    m = refr_material(1.5)
    a, b = ior_pair(m, normalize([1, 0, 1]), [0, 0, -1])
    assert a == 1.5 and b == 1.0


def msample(m, out_, n, xi=(0.3, 0.6, 0.5)):
    i, pdf = (C.c_float * 3)(), C.c_float()
    oracle.lib().lr_oracle_material_sample(C.byref(m), f3(out_), f3(n), f3(xi), i, C.byref(pdf))
    return np.array(i[:], dtype=np.float32), float(pdf.value)


def mbrdf(m, out_, in_, n, pos=(0, 0, 0)):
    r = (C.c_float * 3)()
    oracle.lib().lr_oracle_material_brdf(C.byref(m), f3(out_), f3(in_), f3(n), f3(pos), r)
    return np.array(r[:], dtype=np.float32)


def test_ideal_refraction_brdf_reflecting():
    """ideal_refraction.rs:197-213: ior = INF is a mirror, brdf = 1 / (in . n)."""
    m = refr_material(1e5)
    n = [0, 0, -1]
    out_ = normalize([1, 0, 1])
    on = np.array([0, 0, 1], dtype=np.float32)           # orienting normal of (out_, n)
    for xi3 in (0.0, 0.5, 0.999):                        # whichever branch the Fresnel roulette takes
        in_, _ = msample(m, out_, n, (0.1, 0.2, xi3))
        if xi3 == 0.0:
            assert norm(reflect(out_, on) - in_) < EPS
    in_, _ = msample(m, out_, n, (0.1, 0.2, 0.0))
    brdf = mbrdf(m, out_, in_, n)
    expect = np.ones(3) / float(np.dot(in_, np.array(n, dtype=np.float32)))
    assert norm(expect - brdf) < EPS


def fresnel(a, b, out_, in_, on):
    return float(oracle.lib().lr_oracle_fresnel(a, b, f3(out_), f3(in_), f3(on)))


def test_fresnel_45():
    """ideal_refraction.rs:259-268"""
    on = [0, 0, 1]
    out_ = normalize([1, 0, 1])
    in_ = refract(out_, on, 1.0 / 1.5)
    fr = fresnel(1.0, 1.5, out_, in_, on)
    assert 0.0 < fr <= 1.0


def test_fresnel_sweep():
    """ideal_refraction.rs:270-282"""
    on = [0, 0, 1]
    for i in range(100):
        t = np.float32(i) / np.float32(100.0) * PI / np.float32(2.0)
        out_ = normalize([math.sin(t), 0, math.cos(t)])
        in_ = refract(out_, on, 1.0 / 1.5)
        assert in_ is not None
        fr = fresnel(1.0, 1.5, out_, in_, on)
        assert 0.0 < fr <= 1.0, fr


def test_fresnel_outgoing_sweep():
    """ideal_refraction.rs:284-297"""
    on = [0, 0, 1]
    seen = 0
    for i in range(100):
        t = np.float32(i) / np.float32(100.0) * PI / np.float32(2.0)
        out_ = np.array([math.sin(t), 0, math.cos(t)], dtype=np.float32)
        in_ = refract(out_, on, 1.5 / 1.0)
        if in_ is not None:
            seen += 1
            fr = fresnel(1.5, 1.0, out_, in_, on)
            assert 0.0 < fr <= 1.0, fr
    assert 0 < seen < 100                                 # beyond the critical angle refract() is None


def test_ideal_refraction_sample_unit_length():
    """ideal_refraction.rs:299-312"""
    m = refr_material(1.5)
    for xi3 in (0.0, 0.2, 0.9):
        in_, _ = msample(m, normalize([1, 0, 1]), [0, 0, -1], (0.1, 0.2, xi3))
        assert abs(norm(in_) - 1.0) < EPS
