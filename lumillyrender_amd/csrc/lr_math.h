// lr_math.h -- device math for the gfx950 kernels.
//
// Everything that decides a discrete outcome (hit / miss, which primitive, RR survive, which
// texel) must round exactly like the reference's f32 arithmetic: IEEE + - * / sqrt with NO fused
// multiply-add.  This translation unit is compiled with -ffp-contract=off; only the conservative
// box test (lr_trace.h) re-enables contraction locally.  Transcendentals follow DESIGN.md
// "deterministic math spec": fixed polynomial forms shared with the parity oracle.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define LR_DEV __device__ __forceinline__

namespace lr {

constexpr float kPi  = 3.14159265358979323846264338327950288f;   // constant.rs:1
constexpr float kEps = 1e-3f;                                     // constant.rs:2
constexpr float kInf = 1e5f;                                      // constant.rs:3

struct V3 { float x, y, z; };
LR_DEV V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
LR_DEV V3 v3(float4 a) { return v3(a.x, a.y, a.z); }
LR_DEV V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
LR_DEV V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
LR_DEV V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
LR_DEV V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
LR_DEV V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
LR_DEV V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
LR_DEV V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
LR_DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }            // vector3.rs:77-81
LR_DEV V3 cross(V3 a, V3 b) {                                                         // vector3.rs:83-91
  return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
LR_DEV float sqr_norm(V3 a) { return dot(a, a); }
LR_DEV float norm(V3 a) { return __builtin_sqrtf(sqr_norm(a)); }
LR_DEV V3 normalize(V3 a) { return a / norm(a); }
LR_DEV float fmax_rs(float a, float b) { return __builtin_fmaxf(a, b); }
LR_DEV float fmin_rs(float a, float b) { return __builtin_fminf(a, b); }

// ---- sin / cos (Cephes single-precision forms, both from one reduction) --------------------
LR_DEV void det_sincos(float xx, float* s_out, float* c_out) {
  const float FOPI = 1.27323954473516f;
  const float DP1 = 0.78515625f, DP2 = 2.4187564849853515625e-4f, DP3 = 3.77489497744594108e-8f;
  float x = __builtin_fabsf(xx);
  int j = (int)(FOPI * x);
  float y = (float)j;
  if (j & 1) { j += 1; y += 1.0f; }
  j &= 7;
  bool sneg = xx < 0.0f, cneg = false;
  if (j > 3) { sneg = !sneg; cneg = !cneg; j -= 4; }
  if (j > 1) cneg = !cneg;
  x = ((x - y * DP1) - y * DP2) - y * DP3;
  float z = x * x;
  float pc = ((2.443315711809948E-005f * z - 1.388731625493765E-003f) * z + 4.166664568298827E-002f) * z * z;
  pc = pc - 0.5f * z;
  pc = pc + 1.0f;
  float ps = ((-1.9515295891E-4f * z + 8.3321608736E-3f) * z - 1.6666654611E-1f) * z * x;
  ps = ps + x;
  bool swap = (j == 1 || j == 2);
  float s = swap ? pc : ps, c = swap ? ps : pc;
  *s_out = sneg ? -s : s;
  *c_out = cneg ? -c : c;
}

// The four functions below are written WITHOUT divergent branches: a wave executes every branch some lane takes, and the original
// case analysis (three ranges of atan with two different divisions, three ranges of acos each with its own inlined asin) made the IBL
// lookup of a miss -- which runs at ~15 of 64 lanes -- walk all of them.  Each lane still evaluates exactly the operations of its own
// case, on the same operands and in the same order (the case only SELECTS operands), so the results are the same bits as the branching
// forms the oracle keeps (tests: device == oracle bit for bit on sweeps around every case boundary, test_math_spec_*).
LR_DEV float det_atan(float xx) {
  const float PIO2F = 1.5707963267948966192f, PIO4F = 0.7853981633974483096f;
  float x = __builtin_fabsf(xx);
  const bool big = x > 2.414213562373095f, mid = !big && x > 0.4142135623730950f;
  const float y0 = big ? PIO2F : (mid ? PIO4F : 0.0f);
  // big: -(1 / x) == (-1) / x;  mid: (x - 1) / (x + 1);  else x == x / 1  (one IEEE division instead of two under branches)
  const float num = big ? -1.0f : (mid ? x - 1.0f : x), den = big ? x : (mid ? x + 1.0f : 1.0f);
  x = num / den;
  float z = x * x;
  float p = (((8.05374449538e-2f * z - 1.38776856032E-1f) * z + 1.99777106478E-1f) * z - 3.33329491539E-1f) * z * x + x;
  float y = y0 + p;
  return xx < 0.0f ? -y : y;
}
LR_DEV float det_atan2(float y, float x) {
  const float PIF = 3.141592653589793238f, PIO2F = 1.5707963267948966192f;
  const bool xneg = x < 0.0f, yneg = y < 0.0f;
  const float w = (xneg && !yneg) ? PIF : ((xneg && yneg) ? -PIF : 0.0f);
  float r = w + det_atan(y / x);
  r = y == 0.0f ? (xneg ? PIF : 0.0f) : r;
  r = x == 0.0f ? (yneg ? -PIO2F : (y == 0.0f ? 0.0f : PIO2F)) : r;
  return (x != x || y != y) ? __builtin_nanf("") : r;
}
// asin for |t| <= 0.5 (Cephes asinf below its square-root range, including the tiny-argument shortcut)
LR_DEV float det_asin_half(float t) {
  float a = __builtin_fabsf(t);
  float z = a * a;
  z = ((((4.2163199048E-2f * z + 2.4181311049E-2f) * z + 4.5470025998E-2f) * z + 7.4953002686E-2f) * z + 1.6666752422E-1f) * z * a + a;
  z = a < 1.0e-4f ? a : z;
  return t < 0.0f ? -z : z;
}
LR_DEV float det_acos(float x) {
  const float PIF = 3.141592653589793238f, PIO2F = 1.5707963267948966192f;
  // x < -0.5: pi - 2 asin(sqrt(0.5 (1 + x)));  x > 0.5: 2 asin(sqrt(0.5 (1 - x)));  else pi/2 - asin(x).  1 + x == 1 - |x| for a
  // negative x, the square roots are <= 0.5, so asin never enters its own square-root range: ONE root and ONE polynomial per lane
  const float ax = __builtin_fabsf(x);
  const bool outer = ax > 0.5f;
  const float arg = outer ? __builtin_sqrtf(0.5f * (1.0f - ax)) : x;
  const float r = det_asin_half(arg);
  const float res = outer ? (x > 0.5f ? 2.0f * r : PIF - 2.0f * r) : PIO2F - r;
  return (x != x || ax > 1.0f) ? __builtin_nanf("") : res;
}

// ---- pow / exp through f64 series (rare: Phong / Blinn-Phong lobes, Beer absorption) --------
LR_DEV double det_log2_d(double x) {
  uint64_t b = (uint64_t)__double_as_longlong(x);
  int e = (int)((b >> 52) & 0x7ff);
  if (e == 0) { x = x * 18014398509481984.0; b = (uint64_t)__double_as_longlong(x); e = (int)((b >> 52) & 0x7ff) - 54; }
  e -= 1023;
  b = (b & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
  double m = __longlong_as_double((long long)b);
  if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
  double s = (m - 1.0) / (m + 1.0), s2 = s * s;
  double p = 0.043478260869565216;
  p = p * s2 + 0.047619047619047616;
  p = p * s2 + 0.052631578947368418;
  p = p * s2 + 0.058823529411764705;
  p = p * s2 + 0.066666666666666666;
  p = p * s2 + 0.076923076923076927;
  p = p * s2 + 0.090909090909090912;
  p = p * s2 + 0.1111111111111111;
  p = p * s2 + 0.14285714285714285;
  p = p * s2 + 0.2;
  p = p * s2 + 0.33333333333333331;
  p = p * s2 + 1.0;
  double ln_m = 2.0 * s * p;
  return (double)e + ln_m * 1.4426950408889634;
}
LR_DEV double det_exp2_d(double z) {
  if (z != z) return z;
  if (z > 1025.0) return __longlong_as_double(0x7ff0000000000000LL);
  if (z < -1100.0) return 0.0;
  double n = __builtin_floor(z + 0.5);
  double t = (z - n) * 0.69314718055994529;
  double p = 1.6059043836821613e-10;
  p = p * t + 2.08767569878681e-09;
  p = p * t + 2.505210838544172e-08;
  p = p * t + 2.7557319223985888e-07;
  p = p * t + 2.7557319223985893e-06;
  p = p * t + 2.4801587301587302e-05;
  p = p * t + 0.00019841269841269841;
  p = p * t + 0.0013888888888888889;
  p = p * t + 0.0083333333333333332;
  p = p * t + 0.041666666666666664;
  p = p * t + 0.16666666666666666;
  p = p * t + 0.5;
  p = p * t + 1.0;
  p = p * t + 1.0;
  int ni = (int)n;
  int n1 = ni / 2, n2 = ni - n1;
  double s1 = __longlong_as_double((long long)((uint64_t)(n1 + 1023) << 52));
  double s2 = __longlong_as_double((long long)((uint64_t)(n2 + 1023) << 52));
  return p * s1 * s2;
}
LR_DEV float det_pow(float x, float y) {
  const float FINF = __builtin_huge_valf();
  if (y == 0.0f) return 1.0f;
  if (x != x || y != y) return __builtin_nanf("");
  if (x == 1.0f) return 1.0f;
  bool y_int = __builtin_floorf(y) == y;
  bool y_odd = y_int && __builtin_fabsf(y) < 16777216.0f && (((long long)__builtin_fabsf(y)) & 1);
  bool xneg = (__float_as_uint(x) >> 31) != 0;
  if (x == 0.0f) {
    if (y > 0.0f) return (y_odd && xneg) ? -0.0f : 0.0f;
    return (y_odd && xneg) ? -FINF : FINF;
  }
  float ax = __builtin_fabsf(x);
  if (ax == FINF) {
    float r = y > 0.0f ? FINF : 0.0f;
    return (x < 0.0f && y_odd) ? -r : r;
  }
  if (__builtin_fabsf(y) == FINF) {
    if (ax == 1.0f) return 1.0f;
    return ((ax > 1.0f) == (y > 0.0f)) ? FINF : 0.0f;
  }
  if (x < 0.0f && !y_int) return __builtin_nanf("");
  float r = (float)det_exp2_d((double)y * det_log2_d((double)ax));
  return (x < 0.0f && y_odd) ? -r : r;
}
LR_DEV float det_exp(float x) {
  if (x != x) return x;
  return (float)det_exp2_d((double)x * 1.4426950408889634);
}
// exact f32 remainder for x >= 0, k > 0 (Rust `%`)
LR_DEV float det_fmod_pos(float x, float k) {
  // exact x mod k for 0 <= x < 2^24 and integer-valued k (150, 30, 300, 1 on this path): the quotient
  // estimate may be off by one (it uses a rounded reciprocal), x - q*k is exact either way, and the two
  // fix-ups land on the true remainder, which is always representable.
  if (!(x < 16777216.0f)) return __builtin_fmodf(x, k);
  float q = __builtin_floorf(x * (1.0f / k));
  float r = x - q * k;
  if (r < 0.0f) r = r + k;
  if (r >= k) r = r - k;
  return r;
}
// x mod 1 for x >= 0 (and NaN): the fast path of det_fmod_pos(x, 1) is x - floor(x), exact for every x (>= 2^23: x is an integer and
// the result 0, as fmodf's; inf: NaN, as fmodf's) -- so the IBL lookup needs neither the range test nor the inlined fmodf behind it
LR_DEV float det_fmod1_pos(float x) { return x - __builtin_floorf(x); }
// 1.0f / d, correctly rounded, in five instructions instead of the compiler's ten (v_div_scale x2, v_rcp, 4 fma,
// mul, v_div_fmas, v_div_fixup): v_rcp_f32 (1 ulp) and two Newton steps with fused residuals.  Bit-equal to the
// IEEE quotient for every float with 2^-126 <= |d| < 2^126 (biased exponent 1..252) -- checked EXHAUSTIVELY by
// lr_selftest_rcp (tests/test_gpu_math.py::test_fast_reciprocal_is_ieee_exact).  The one user is the triangle
// test's 1/det: it discards the value when |det| < 1e-3, and |det| <= |e1| |e2| |d| stays below 2^126 because
// lr_scene_create refuses triangles with |e1| |e2| >= 2^120 (edges of ~1e18 units).  A guarded form with the
// compiler's sequence as a wave-uniform fallback inside the loop cost 9 % of the frame, so the guard sits on the host.
LR_DEV float rcp_exact_mid(float d) {
  float r = __builtin_amdgcn_rcpf(d);
  float e = __builtin_fmaf(-d, r, 1.0f);
  r = __builtin_fmaf(e, r, r);
  e = __builtin_fmaf(-d, r, 1.0f);
  return __builtin_fmaf(e, r, r);
}
LR_DEV float rcp_exact_mid3(float d) {             // one more step (diagnostic variant of the exhaustive test)
  float r = rcp_exact_mid(d);
  float e = __builtin_fmaf(-d, r, 1.0f);
  return __builtin_fmaf(e, r, r);
}
LR_DEV float det_powi(float a, int b) {           // compiler-rt __powisf2
  bool recip = b < 0;
  float r = 1.0f;
  while (true) {
    if (b & 1) r = r * a;
    b /= 2;
    if (b == 0) break;
    a = a * a;
  }
  return recip ? 1.0f / r : r;
}

// ---- counter-based RNG: pcg4d over (pixel, sample, block, seed) ------------------------------
struct Draw4 { float v[4]; };
// a * b + c in 32 bits.  Left to itself the compiler fuses it into v_mad_u64_u32 (a 64-bit sum nobody reads, on a register PAIR);
// LO = true spells it v_mul_lo_u32 + v_add_u32.  Same integers either way.  Which one is faster is a matter of registers, not of
// issue slots (round 4, interleaved A/B): the pt-direct tree kernel drops from 13 to 4 spilled VGPRs with LO (48 -> 32 B of
// scratch; config 5 3950 -> 4017, +1.7 %), the kernels without spills lose 1.3-1.6 % with it (config 4 4109 -> 4057, configs[1]
// 6110 -> 6015) -- so the caller chooses: shade_vertex_core takes LO for k_path_tree<., true> only (RngLo<LaneStateT<false>>); the camera
// block of path_spare_batch and every other caller keep the compiler's spelling (rng_block<false>), which is what was measured.
template <bool LO>
LR_DEV uint32_t mad32(uint32_t a, uint32_t b, uint32_t c) {
  if constexpr (LO) {
    uint32_t p;
    asm("v_mul_lo_u32 %0, %1, %2" : "=v"(p) : "v"(a), "v"(b));
    return p + c;
  } else {
    return a * b + c;
  }
}
template <bool LO = false>
LR_DEV Draw4 rng_block(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t block) {
  uint32_t x = pixel, y = sample, z = block, w = seed;
  x = mad32<LO>(x, 1664525u, 1013904223u); y = mad32<LO>(y, 1664525u, 1013904223u);
  z = mad32<LO>(z, 1664525u, 1013904223u); w = mad32<LO>(w, 1664525u, 1013904223u);
  x = mad32<LO>(y, w, x); y = mad32<LO>(z, x, y); z = mad32<LO>(x, y, z); w = mad32<LO>(y, z, w);
  x ^= x >> 16; y ^= y >> 16; z ^= z >> 16; w ^= w >> 16;
  x = mad32<LO>(y, w, x); y = mad32<LO>(z, x, y); z = mad32<LO>(x, y, z); w = mad32<LO>(y, z, w);
  Draw4 d;
  d.v[0] = (float)(x >> 8) * 5.9604644775390625e-08f;
  d.v[1] = (float)(y >> 8) * 5.9604644775390625e-08f;
  d.v[2] = (float)(z >> 8) * 5.9604644775390625e-08f;
  d.v[3] = (float)(w >> 8) * 5.9604644775390625e-08f;
  return d;
}

}  // namespace lr
