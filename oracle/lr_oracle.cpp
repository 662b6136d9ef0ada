// lr_oracle.cpp -- CPU ORACLE: a restatement of LumillyRender's per-pixel sampling loop.
//
// THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may load liboracle.so.  Nothing under lumillyrender_amd/ includes,
// links or calls anything in this directory.
//
// PARITY STATUS: "parity unpinned" for everything except the items below.  The reference is Rust
// (nightly 2018, no toolchain in this image, crates not vendored) and draws every random number
// from an OS-seeded generator, so it can be neither built nor reproduced here.  Pinned against the
// reference's own known-answer tests (tests/test_oracle_kat.py):
//   * triangle.rs:157-235  (4 vectors: Moller-Trumbore vs 3-cross, front/back/near)
//   * util.rs:49-81        (reflect, total internal reflection, Snell at 30 degrees)
//   * material/ideal_refraction.rs:167-312 (ior_pair, mirror limit, Fresnel range, unit length)
// Everything else follows the reference line by line (each function cites file:line) but has no
// reference-side vector to check against.
//
// Two things are necessarily NOT the reference's:
//   * RNG: rand::random::<f32>() (30 call sites) is replaced by a counter-based generator keyed by
//     (seed, pixel, sample, draw-slot) -- see rng_block().  Draw ORDER per path vertex follows the
//     reference (scene.rs:173-193): [RR] -> [light pick, u, v] -> [bsdf r1, r2 (, r3)].
//   * libm: sin/cos/acos/atan2/powf/exp are restated with fixed polynomial forms (only IEEE
//     + - * / sqrt and integer ops) so that the HIP kernels can reproduce them bit for bit.
//     They are accurate to ~1-2 ulp (tests compare them with numpy).
//
// Closest-hit semantics (bvh.rs:20-25,130-141): a primitive is a candidate only if ITS OWN exact box passes
// aabb.rs:74-92 (Leaf::may_intersect; the slab test is seeded with [-1e5, 1e5], constant.rs:3); the result is the
// minimum distance over the candidates whose own test accepts the ray.  The boxes of inner nodes never reject what
// the leaf's box accepts (f32 rounding is monotone in the box planes; tests/test_oracle_properties.py checks it on
// edge-aimed rays), so this is tree-independent up to EXACT distance ties, where the reference keeps the first
// minimum in the candidate order of its tree (bvh.rs:38-45,131-141).
//   mode 3 (OWNBOX)      = that definition: every primitive, own exact box, own test; an exact tie goes to the primitive that comes
//                          first in the reference's candidate order (the depth-first leaf order of its SAH tree with stable sorts; the
//                          tree is built for that order only).  What the HIP path implements with a host-built tree since round 6.
//   mode 1 (BVH)         = the reference line by line: SAH tree + collect-all-candidates walk, first minimum in candidate order
//                          (pad = 0).  Equal to mode 3 on every ray (mode 6 audits it); mode 4 (OWNBOX_TREE) is its alias at pad 0
//   mode 7 (OWNBOX_INDEX) / 8 (OWNBOX_TREE_INDEX) = the definition / the tree walk at pad 0 with exact ties to the lowest primitive INDEX:
//                          what the HIP path gives on a device-built tree (no reference order to follow)
//   mode 0 (BRUTE)       = minimum over ALL primitives, no box at all, ties to the lowest index (rounds 1-5 defined the device by it;
//                          kept as the checker of lr_selftest_brute and of the box-free traversal tests)
//   mode 2 (BVH_ORDERED) = near-first walk with early out over padded boxes (= mode 0)
//   mode 5 (OWNBOX_ORDERED) = mode 2 with the primitive's own exact box behind its test and mode 3's tie rule (= mode 3; needs
//                          pad > 0 for the inner boxes; the "optimized" CPU baseline)
//   mode 6 (BVH_AUDIT)   = mode 1 at pad 0, and every query is also answered by the definition (mode 3): LrOracleStats counts the
//                          queries that differ by an exact tie (tie_flips) and in any other way (order_dependent); both must stay 0
//
// Build: see oracle/Makefile  (g++ -O2 -ffp-contract=off, no fast-math).

#include <cmath>
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <thread>
#include <atomic>
#include <limits>
#include <chrono>

#include "../include/lumilly_hip.h"   // POD boundary structs only (no product code)

namespace {

// ------------------------------------------------------------------------------------------
// constants  (src/constant.rs:1-3)
// ------------------------------------------------------------------------------------------
const float PI  = 3.14159265358979323846264338327950288f;
const float EPS = 1e-3f;
const float INF = 1e5f;

// ------------------------------------------------------------------------------------------
// Vector3  (src/math/vector3.rs:7-162, src/math/traits.rs:25-42)
// ------------------------------------------------------------------------------------------
struct V3 { float x, y, z; };
inline V3 v3(float x, float y, float z) { V3 r = {x, y, z}; return r; }
inline V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
inline V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }   // vector3.rs:77-81
inline V3 cross(V3 a, V3 b) {                                                  // vector3.rs:83-91
  return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline float sqr_norm(V3 a) { return dot(a, a); }                              // traits.rs:32-34
inline float norm(V3 a) { return std::sqrt(sqr_norm(a)); }                     // traits.rs:28-30
inline V3 normalize(V3 a) { return a / norm(a); }                              // traits.rs:40-42
inline float comp(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
inline float fmax_rs(float a, float b) { return std::fmax(a, b); }             // f32::max
inline float fmin_rs(float a, float b) { return std::fmin(a, b); }             // f32::min

// ------------------------------------------------------------------------------------------
// deterministic libm replacements (DESIGN.md "deterministic math spec")
// ------------------------------------------------------------------------------------------
// sin/cos: Cephes single-precision forms; |x| <= ~8192 keeps the 3-term reduction exact enough.
void det_sincos(float xx, float* s_out, float* c_out) {
  const float FOPI = 1.27323954473516f;
  const float DP1 = 0.78515625f, DP2 = 2.4187564849853515625e-4f, DP3 = 3.77489497744594108e-8f;
  float x = std::fabs(xx);
  int j = (int)(FOPI * x);
  float y = (float)j;
  if (j & 1) { j += 1; y += 1.0f; }
  j &= 7;
  float ssign = xx < 0.0f ? -1.0f : 1.0f;
  float csign = 1.0f;
  if (j > 3) { ssign = -ssign; csign = -csign; j -= 4; }
  if (j > 1) csign = -csign;
  x = ((x - y * DP1) - y * DP2) - y * DP3;
  float z = x * x;
  float pc = ((2.443315711809948E-005f * z - 1.388731625493765E-003f) * z + 4.166664568298827E-002f) * z * z;
  pc = pc - 0.5f * z;
  pc = pc + 1.0f;
  float ps = ((-1.9515295891E-4f * z + 8.3321608736E-3f) * z - 1.6666654611E-1f) * z * x;
  ps = ps + x;
  float s, c;
  if (j == 1 || j == 2) { s = pc; c = ps; } else { s = ps; c = pc; }
  *s_out = ssign < 0.0f ? -s : s;
  *c_out = csign < 0.0f ? -c : c;
}
inline float det_sin(float x) { float s, c; det_sincos(x, &s, &c); return s; }
inline float det_cos(float x) { float s, c; det_sincos(x, &s, &c); return c; }

float det_atan(float xx) {
  const float PIO2F = 1.5707963267948966192f, PIO4F = 0.7853981633974483096f;
  float x = std::fabs(xx), y;
  if (x > 2.414213562373095f) { y = PIO2F; x = -(1.0f / x); }
  else if (x > 0.4142135623730950f) { y = PIO4F; x = (x - 1.0f) / (x + 1.0f); }
  else y = 0.0f;
  float z = x * x;
  float p = (((8.05374449538e-2f * z - 1.38776856032E-1f) * z + 1.99777106478E-1f) * z - 3.33329491539E-1f) * z * x + x;
  y = y + p;
  return xx < 0.0f ? -y : y;
}
float det_atan2(float y, float x) {
  const float PIF = 3.141592653589793238f, PIO2F = 1.5707963267948966192f;
  if (x != x || y != y) return std::numeric_limits<float>::quiet_NaN();
  int code = 0;
  if (x < 0.0f) code = 2;
  if (y < 0.0f) code |= 1;
  if (x == 0.0f) {
    if (code & 1) return -PIO2F;
    if (y == 0.0f) return 0.0f;
    return PIO2F;
  }
  if (y == 0.0f) return (code & 2) ? PIF : 0.0f;
  float w = (code == 2) ? PIF : (code == 3 ? -PIF : 0.0f);
  return w + det_atan(y / x);
}
float det_asin(float xx) {
  const float PIO2F = 1.5707963267948966192f;
  float a = std::fabs(xx);
  if (a > 1.0f || a != a) return std::numeric_limits<float>::quiet_NaN();
  if (a < 1.0e-4f) return xx;
  float x, z; int flag;
  if (a > 0.5f) { z = 0.5f * (1.0f - a); x = std::sqrt(z); flag = 1; }
  else { x = a; z = x * x; flag = 0; }
  z = ((((4.2163199048E-2f * z + 2.4181311049E-2f) * z + 4.5470025998E-2f) * z + 7.4953002686E-2f) * z + 1.6666752422E-1f) * z * x + x;
  if (flag) { z = z + z; z = PIO2F - z; }
  return xx < 0.0f ? -z : z;
}
float det_acos(float x) {
  const float PIF = 3.141592653589793238f, PIO2F = 1.5707963267948966192f;
  if (x != x || x < -1.0f || x > 1.0f) return std::numeric_limits<float>::quiet_NaN();
  if (x < -0.5f) return PIF - 2.0f * det_asin(std::sqrt(0.5f * (1.0f + x)));
  if (x > 0.5f) return 2.0f * det_asin(std::sqrt(0.5f * (1.0f - x)));
  return PIO2F - det_asin(x);
}

// pow/exp go through double with fixed series (exact same operation list on the device)
double det_log2_d(double x) {   // x > 0, finite
  uint64_t b; std::memcpy(&b, &x, 8);
  int e = (int)((b >> 52) & 0x7ff);
  if (e == 0) { x = x * 18014398509481984.0; std::memcpy(&b, &x, 8); e = (int)((b >> 52) & 0x7ff) - 54; }
  e -= 1023;
  b = (b & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
  double m; std::memcpy(&m, &b, 8);
  if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
  double s = (m - 1.0) / (m + 1.0), s2 = s * s;
  double p = 0.043478260869565216;            // 1/23
  p = p * s2 + 0.047619047619047616;          // 1/21
  p = p * s2 + 0.052631578947368418;          // 1/19
  p = p * s2 + 0.058823529411764705;          // 1/17
  p = p * s2 + 0.066666666666666666;          // 1/15
  p = p * s2 + 0.076923076923076927;          // 1/13
  p = p * s2 + 0.090909090909090912;          // 1/11
  p = p * s2 + 0.1111111111111111;            // 1/9
  p = p * s2 + 0.14285714285714285;           // 1/7
  p = p * s2 + 0.2;                           // 1/5
  p = p * s2 + 0.33333333333333331;           // 1/3
  p = p * s2 + 1.0;
  double ln_m = 2.0 * s * p;
  return (double)e + ln_m * 1.4426950408889634;
}
double det_exp2_d(double z) {
  if (z != z) return z;
  if (z > 1025.0) return std::numeric_limits<double>::infinity();
  if (z < -1100.0) return 0.0;
  double n = std::floor(z + 0.5);
  double t = (z - n) * 0.69314718055994529;
  double p = 1.6059043836821613e-10;          // 1/13!
  p = p * t + 2.08767569878681e-09;           // 1/12!
  p = p * t + 2.505210838544172e-08;          // 1/11!
  p = p * t + 2.7557319223985888e-07;         // 1/10!
  p = p * t + 2.7557319223985893e-06;         // 1/9!
  p = p * t + 2.4801587301587302e-05;         // 1/8!
  p = p * t + 0.00019841269841269841;         // 1/7!
  p = p * t + 0.0013888888888888889;          // 1/6!
  p = p * t + 0.0083333333333333332;          // 1/5!
  p = p * t + 0.041666666666666664;           // 1/4!
  p = p * t + 0.16666666666666666;            // 1/3!
  p = p * t + 0.5;
  p = p * t + 1.0;
  p = p * t + 1.0;
  int ni = (int)n;
  // scale by 2^ni in two exact steps (keeps subnormal results correctly rounded once)
  int n1 = ni / 2, n2 = ni - n1;
  uint64_t b1 = (uint64_t)(n1 + 1023) << 52, b2 = (uint64_t)(n2 + 1023) << 52;
  double s1, s2; std::memcpy(&s1, &b1, 8); std::memcpy(&s2, &b2, 8);
  return p * s1 * s2;
}
// f32::powf semantics for the cases the path can reach
float det_pow(float x, float y) {
  if (y == 0.0f) return 1.0f;
  if (x != x || y != y) return std::numeric_limits<float>::quiet_NaN();
  if (x == 1.0f) return 1.0f;
  bool y_int = std::floor(y) == y;
  bool y_odd = y_int && std::fabs(y) < 16777216.0f && (((long long)std::fabs(y)) & 1);
  if (x == 0.0f) {
    if (y > 0.0f) return (y_odd && std::signbit(x)) ? -0.0f : 0.0f;
    return (y_odd && std::signbit(x)) ? -std::numeric_limits<float>::infinity() : std::numeric_limits<float>::infinity();
  }
  float ax = std::fabs(x);
  if (std::isinf(ax)) {
    float r = y > 0.0f ? std::numeric_limits<float>::infinity() : 0.0f;
    return (x < 0.0f && y_odd) ? -r : r;
  }
  if (std::isinf(y)) {
    if (ax == 1.0f) return 1.0f;
    return ((ax > 1.0f) == (y > 0.0f)) ? std::numeric_limits<float>::infinity() : 0.0f;
  }
  if (x < 0.0f && !y_int) return std::numeric_limits<float>::quiet_NaN();
  float r = (float)det_exp2_d((double)y * det_log2_d((double)ax));
  return (x < 0.0f && y_odd) ? -r : r;
}
float det_exp(float x) {
  if (x != x) return x;
  return (float)det_exp2_d((double)x * 1.4426950408889634);
}
// exact f32 remainder x % k for x >= 0, k > 0 (Rust's `%` on f32 is fmod)
float det_fmod_pos(float x, float k) {
  // exact x mod k for 0 <= x < 2^24 and integer-valued k (150, 30, 300, 1 on this path): the quotient
  // estimate may be off by one (it uses a rounded reciprocal), x - q*k is exact either way, and the two
  // fix-ups land on the true remainder, which is always representable.
  if (!(x < 16777216.0f)) return std::fmod(x, k);
  float q = std::floor(x * (1.0f / k));
  float r = x - q * k;
  if (r < 0.0f) r = r + k;
  if (r >= k) r = r - k;
  return r;
}
// f32::powi via repeated squaring (compiler-rt __powisf2)
float det_powi(float a, int b) {
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

// ------------------------------------------------------------------------------------------
// counter-based RNG replacing rand::random::<f32>()
//   pcg4d (Jarzynski & Olano, "Hash Functions for GPU Rendering", JCGT 2020) over the key
//   (pixel index, sample index, draw block, seed); 4 draws per block, 24-bit mantissa in [0,1).
//   block 0 = camera (sensor u, v, aperture u, v); vertex d uses block 1+2d = (rr, pick, light u,
//   light v) and block 2+2d = (bsdf r1, bsdf r2, bsdf r3, spare).
// ------------------------------------------------------------------------------------------
struct U4 { uint32_t x, y, z, w; };
U4 pcg4d(U4 v) {
  v.x = v.x * 1664525u + 1013904223u; v.y = v.y * 1664525u + 1013904223u;
  v.z = v.z * 1664525u + 1013904223u; v.w = v.w * 1664525u + 1013904223u;
  v.x += v.y * v.w; v.y += v.z * v.x; v.z += v.x * v.y; v.w += v.y * v.z;
  v.x ^= v.x >> 16; v.y ^= v.y >> 16; v.z ^= v.z >> 16; v.w ^= v.w >> 16;
  v.x += v.y * v.w; v.y += v.z * v.x; v.z += v.x * v.y; v.w += v.y * v.z;
  return v;
}
struct Draw4 { float v[4]; };
Draw4 rng_block(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t block) {
  U4 k = {pixel, sample, block, seed};
  U4 r = pcg4d(k);
  Draw4 d;
  d.v[0] = (float)(r.x >> 8) * 5.9604644775390625e-08f;
  d.v[1] = (float)(r.y >> 8) * 5.9604644775390625e-08f;
  d.v[2] = (float)(r.z >> 8) * 5.9604644775390625e-08f;
  d.v[3] = (float)(r.w >> 8) * 5.9604644775390625e-08f;
  return d;
}
// per-path draw context: where in the stream the path currently is
struct Rng {
  uint32_t seed, pixel, sample;
  Draw4 block(uint32_t b) const { return rng_block(seed, pixel, sample, b); }
};

// ------------------------------------------------------------------------------------------
// Ray / Sample / Intersection  (ray.rs:3-6, sample.rs:1-4, intersection.rs:5-10)
// ------------------------------------------------------------------------------------------
struct Ray { V3 origin, direction; };
struct Intersection { V3 position; float distance; V3 normal; int material; int prim; };

// ------------------------------------------------------------------------------------------
// util.rs
// ------------------------------------------------------------------------------------------
void orthonormal_basis(V3 w, V3* t_out, V3* b_out) {            // util.rs:12-21
  V3 a = std::fabs(w.x) > EPS ? v3(0.0f, 1.0f, 0.0f) : v3(1.0f, 0.0f, 0.0f);
  V3 tangent = normalize(cross(a, w));
  V3 binormal = cross(w, tangent);
  *t_out = tangent; *b_out = binormal;
}
V3 reflect(V3 self, V3 normal) {                                // util.rs:30-32
  return -self + normal * (dot(self, normal) * 2.0f);
}
bool refract(V3 self, V3 normal, float from_per_to_ior, V3* out) {   // util.rs:34-42
  float dn = dot(self, normal);
  float cos2theta = 1.0f - det_powi(from_per_to_ior, 2) * (1.0f - det_powi(dn, 2));
  if (cos2theta > 0.0f) {
    *out = -self * from_per_to_ior - normal * (from_per_to_ior * -dn + std::sqrt(cos2theta));
    return true;
  }
  return false;
}
V3 hemisphere_cos_importance(float xi1, float xi2) {            // util.rs:87-96
  float r1 = 2.0f * PI * xi1;
  float r2 = xi2;
  float r2s = std::sqrt(r2);
  return v3(det_cos(r1) * r2s, det_sin(r1) * r2s, std::sqrt(1.0f - r2));
}
V3 sphere_uniform(float xi1, float xi2) {                       // util.rs:108-116
  float r1 = 2.0f * PI * xi1;
  float r2 = xi2 * 2.0f - 1.0f;
  float r2s = std::sqrt(1.0f - r2 * r2);
  return v3(det_cos(r1) * r2s, det_sin(r1) * r2s, r2);
}

// ------------------------------------------------------------------------------------------
// materials  (material/*.rs)
// ------------------------------------------------------------------------------------------
V3 orienting_normal(V3 out_, V3 normal) {                       // lambert.rs:14-21 (same in all five)
  if (dot(normal, out_) < 0.0f) return normal * -1.0f;
  return normal;
}
float signed_mod(float base, float module) {                    // lambert.rs:58-64
  if (base > 0.0f) return det_fmod_pos(base, module);
  return module - det_fmod_pos(-base, module);
}
V3 checker(float u, float v) {                                  // lambert.rs:66-90
  const float lw = 2.0f, li = 150.0f, sw = 1.0f, si = 30.0f, cw = 150.0f, ci = 300.0f;
  float lu = signed_mod(u, li), lv = signed_mod(v, li);
  float su = signed_mod(u, si), sv = signed_mod(v, si);
  float cu = signed_mod(u, ci), cv = signed_mod(v, ci);
  if (lu < lw || lv < lw) return v3(0.5f, 0.5f, 0.5f);
  else if (su < sw || sv < sw) return v3(0.6f, 0.6f, 0.6f);
  else if ((cu < cw || cv < cw) && !(cu < cw && cv < cw)) return v3(0.8f, 0.8f, 0.8f);
  else return v3(1.0f, 1.0f, 1.0f);
}
inline V3 mcolor(const LrMaterial& m) { return v3(m.color[0], m.color[1], m.color[2]); }

V3 material_emission(const LrMaterial& m) {                     // lambert.rs:23-25; others return zero
  if (m.type == LR_MAT_LAMBERT) return v3(m.emission[0], m.emission[1], m.emission[2]);
  return v3(0.0f, 0.0f, 0.0f);
}
float material_weight(const LrMaterial& m) {                    // lambert.rs:27-30, phong.rs:30-35, ...
  return fmax_rs(fmax_rs(m.color[0], m.color[1]), m.color[2]);
}
// GGX helpers (ggx.rs:18-48)
float ggx_alpha(const LrMaterial& m) { return m.param[0] * m.param[0]; }
float ggx_g(const LrMaterial& m, V3 v, V3 n) {                  // ggx.rs:27-32
  float a2 = ggx_alpha(m) * ggx_alpha(m);
  float c = dot(v, n);
  float tan = 1.0f / (c * c) - 1.0f;
  return 2.0f / (1.0f + std::sqrt(1.0f + a2 * tan * tan));
}
float ggx_ndf(const LrMaterial& m, V3 mm, V3 n) {               // ggx.rs:34-39
  float a2 = ggx_alpha(m) * ggx_alpha(m);
  float mdn = dot(mm, n);
  float x = (a2 - 1.0f) * mdn * mdn + 1.0f;
  return a2 / (PI * x * x);
}
float ggx_fresnel(const LrMaterial& m, V3 in_, V3 mm) {         // ggx.rs:41-47
  float ior = m.param[1];
  float nnn = 1.0f - ior, nnp = 1.0f + ior;
  float f_0 = (nnn * nnn) / (nnp * nnp);
  float c = dot(in_, mm);
  return f_0 + (1.0f - f_0) * det_powi(1.0f - c, 5);
}
// IdealRefraction helpers (ideal_refraction.rs:116-160)
void ior_pair(const LrMaterial& m, V3 out_, V3 n, float* from_ior, float* to_ior) {   // :117-135
  float ior_v = 1.0f, ior = m.param[0];
  if (dot(out_, n) > 0.0f) { *from_ior = ior_v; *to_ior = ior; }
  else { *from_ior = ior; *to_ior = ior_v; }
}
float fresnel_exact(float from_ior, float to_ior, V3 out_, V3 in_, V3 on) {           // :137-149
  float cos1 = dot(out_, on);
  float cos2 = dot(in_, -on);
  float n1 = from_ior, n2 = to_ior;
  float rs = det_powi((n1 * cos1 - n2 * cos2) / (n1 * cos1 + n2 * cos2), 2);
  float rp = det_powi((n1 * cos2 - n2 * cos1) / (n1 * cos2 + n2 * cos1), 2);
  return (rs + rp) / 2.0f;
}

V3 material_brdf(const LrMaterial& m, V3 out_, V3 in_, V3 n, V3 pos) {
  switch (m.type) {
    case LR_MAT_LAMBERT:                                        // lambert.rs:32-35
      return mcolor(m) * checker(pos.x, pos.z) / PI;
    case LR_MAT_PHONG: {                                        // phong.rs:37-45
      V3 on = orienting_normal(out_, n);
      if (dot(in_, on) <= 0.0f) return v3(0, 0, 0);
      V3 r = reflect(out_, on);
      float c = dot(r, in_);
      float a = m.param[0];
      return mcolor(m) * ((a + 2.0f) / (2.0f * PI) * det_pow(c, a));
    }
    case LR_MAT_BLINN_PHONG: {                                  // blinn_phong.rs:37-47
      V3 on = orienting_normal(out_, n);
      if (dot(in_, on) <= 0.0f) return v3(0, 0, 0);
      V3 h = normalize(in_ + out_);
      float c = dot(h, on);
      float a = m.param[0];
      return mcolor(m) * ((a + 2.0f) * (a + 4.0f) / (8.0f * PI * (det_pow(2.0f, -a / 2.0f) + a)) * det_pow(c, a));
    }
    case LR_MAT_GGX: {                                          // ggx.rs:71-85
      V3 on = orienting_normal(out_, n);
      if (dot(in_, on) <= 0.0f) return v3(0, 0, 0);
      V3 h = normalize(in_ + out_);
      float f = ggx_fresnel(m, in_, h);
      float g = ggx_g(m, in_, on) * ggx_g(m, out_, on);         // gaf_smith ggx.rs:23-25
      float d = ggx_ndf(m, h, on);
      return mcolor(m) * f * g * d / (4.0f * dot(in_, on) * dot(out_, on));
    }
    case LR_MAT_IDEAL_REFRACTION: {                             // ideal_refraction.rs:39-66
      V3 on = orienting_normal(out_, n);
      float from_ior, to_ior; ior_pair(m, out_, n, &from_ior, &to_ior);
      float from_per_to_ior = from_ior / to_ior;
      V3 r;
      if (refract(out_, on, from_per_to_ior, &r)) {
        float fr = fresnel_exact(from_ior, to_ior, out_, r, on);
        if (dot(in_, on) > 0.0f) return mcolor(m) * 1.0f / dot(in_, n) * fr;
        float ft = (1.0f - fr) * det_powi(to_ior / from_ior, 2);
        return mcolor(m) * 1.0f / dot(in_, n) * ft;
      }
      return mcolor(m) * 1.0f / dot(in_, n);
    }
  }
  return v3(0, 0, 0);
}

// xi = (r1, r2, r3) from the vertex's bsdf block
void material_sample(const LrMaterial& m, V3 out_, V3 n, const float* xi, V3* in_out, float* pdf_out) {
  switch (m.type) {
    case LR_MAT_LAMBERT: {                                      // lambert.rs:37-55
      V3 on = orienting_normal(out_, n);
      V3 w = on, u, v; orthonormal_basis(w, &u, &v);
      V3 s = hemisphere_cos_importance(xi[0], xi[1]);
      V3 in_ = u * s.x + v * s.y + w * s.z;
      float cos_term = dot(in_, n);
      *in_out = in_; *pdf_out = cos_term / PI;
      return;
    }
    case LR_MAT_PHONG: {                                        // phong.rs:47-68
      V3 on = orienting_normal(out_, n);
      float a = m.param[0];
      V3 r = reflect(out_, on);
      V3 w = r, u, v; orthonormal_basis(w, &u, &v);
      float r1 = 2.0f * PI * xi[0];
      float r2 = xi[1];
      float t = det_pow(r2, 1.0f / (a + 2.0f));
      float ts = std::sqrt(1.0f - t * t);
      V3 in_ = u * det_cos(r1) * ts + v * det_sin(r1) * ts + w * t;
      float c = dot(r, in_);
      *in_out = in_; *pdf_out = (a + 2.0f) / (2.0f * PI) * det_pow(c, a);
      return;
    }
    case LR_MAT_BLINN_PHONG: {                                  // blinn_phong.rs:49-72
      V3 on = orienting_normal(out_, n);
      float a = m.param[0];
      V3 w = on, u, v; orthonormal_basis(w, &u, &v);
      float r1 = 2.0f * PI * xi[0];
      float r2 = xi[1];
      float t = det_pow(r2, 1.0f / (a + 2.0f));
      float ts = std::sqrt(1.0f - t * t);
      V3 h = u * det_cos(r1) * ts + v * det_sin(r1) * ts + w * t;
      V3 in_ = h * (2.0f * dot(out_, h)) - out_;
      float c = dot(on, h);
      *in_out = in_; *pdf_out = (a + 2.0f) / (2.0f * PI) * det_pow(c, a);
      return;
    }
    case LR_MAT_GGX: {                                          // ggx.rs:87-113
      V3 on = orienting_normal(out_, n);
      V3 w = on, u, v; orthonormal_basis(w, &u, &v);
      float r1 = 2.0f * PI * xi[0];
      float r2 = xi[1];
      float tan = ggx_alpha(m) * std::sqrt(r2 / (1.0f - r2));
      float x = 1.0f + tan * tan;
      float c = 1.0f / std::sqrt(x);
      float s = tan / std::sqrt(x);
      V3 h = u * det_cos(r1) * s + v * det_sin(r1) * s + w * c;
      float o_h = dot(out_, h);
      V3 in_ = h * (2.0f * o_h) - out_;
      float jacobian = 1.0f / (4.0f * o_h);
      *in_out = in_; *pdf_out = ggx_ndf(m, h, on) * dot(h, on) * jacobian;
      return;
    }
    case LR_MAT_IDEAL_REFRACTION: {                             // ideal_refraction.rs:68-104
      float from_ior, to_ior; ior_pair(m, out_, n, &from_ior, &to_ior);
      float from_per_to_ior = from_ior / to_ior;
      V3 on = orienting_normal(out_, n);
      V3 r;
      if (refract(out_, on, from_per_to_ior, &r)) {
        float fr = fresnel_exact(from_ior, to_ior, out_, r, on);
        float rr_prob = fr;
        if (xi[2] < rr_prob) { *in_out = reflect(out_, on); *pdf_out = 1.0f * rr_prob; }
        else { *in_out = r; *pdf_out = 1.0f * (1.0f - rr_prob); }
        return;
      }
      *in_out = reflect(out_, on); *pdf_out = 1.0f;
      return;
    }
  }
  *in_out = v3(0, 0, 0); *pdf_out = 0.0f;
}

V3 material_coef(const LrMaterial& m, V3 out_, V3 n, float fly_distance) {   // traits.rs:20-22; ideal_refraction.rs:106-113
  if (m.type == LR_MAT_IDEAL_REFRACTION && dot(out_, n) < 0.0f) {
    V3 v = -(v3(1.0f, 1.0f, 1.0f) - mcolor(m)) * m.param[1] * fly_distance;
    return v3(det_exp(v.x), det_exp(v.y), det_exp(v.z));
  }
  return v3(1.0f, 1.0f, 1.0f);
}

// ------------------------------------------------------------------------------------------
// primitives  (triangle.rs, sphere.rs)
// ------------------------------------------------------------------------------------------
struct Prim {
  int type, material;
  V3 p0, p1, p2;      // triangle
  V3 normal;          // triangle.rs:36
  V3 centre; float radius;   // sphere
  float area;         // triangle.rs:37 / sphere.rs:25
  V3 bmin, bmax, bcentre;    // triangle.rs:102-118 / sphere.rs:31-38
};

bool triangle_intersect_mt(const Prim& t, const Ray& ray, Intersection* out) {   // triangle.rs:69-100
  V3 e1 = t.p1 - t.p0;
  V3 e2 = t.p2 - t.p0;
  V3 pv = cross(ray.direction, e2);
  float det = dot(e1, pv);
  if (std::fabs(det) < EPS) return false;
  float invdet = 1.0f / det;
  V3 tv = ray.origin - t.p0;
  float u = dot(tv, pv) * invdet;
  if (u < 0.0f || u > 1.0f) return false;
  V3 qv = cross(tv, e1);
  float v = dot(ray.direction, qv) * invdet;
  if (v < 0.0f || u + v > 1.0f) return false;
  float tt = dot(e2, qv) * invdet;
  if (tt < EPS) return false;
  out->distance = tt;
  out->normal = t.normal;
  out->position = ray.origin + ray.direction * tt;
  out->material = t.material;
  return true;
}
bool triangle_intersect_3c(const Prim& t, const Ray& ray, Intersection* out) {   // triangle.rs:42-67 (test-only)
  float dn = dot(ray.direction, t.normal);
  float tt = dot(t.p0 - ray.origin, t.normal) / dn;
  if (tt < EPS) return false;
  V3 p = ray.origin + ray.direction * tt;
  V3 c0 = cross(t.p1 - t.p0, p - t.p0);
  if (dot(c0, t.normal) < 0.0f) return false;
  V3 c1 = cross(t.p2 - t.p1, p - t.p1);
  if (dot(c1, t.normal) < 0.0f) return false;
  V3 c2 = cross(t.p0 - t.p2, p - t.p2);
  if (dot(c2, t.normal) < 0.0f) return false;
  out->distance = tt; out->normal = t.normal; out->position = p; out->material = t.material;
  return true;
}
bool sphere_intersect(const Prim& s, const Ray& ray, Intersection* out) {        // sphere.rs:42-63
  V3 co = ray.origin - s.centre;
  float cod = dot(co, ray.direction);
  float det = cod * cod - sqr_norm(co) + s.radius * s.radius;
  if (det <= 0.0f) return false;
  float t1 = -cod - std::sqrt(det);
  float t2 = -cod + std::sqrt(det);
  if (t1 < EPS && t2 < EPS) return false;
  float distance = t1 > EPS ? t1 : t2;
  V3 position = ray.origin + ray.direction * distance;
  V3 outer_normal = normalize(position - s.centre);
  out->distance = distance; out->position = position; out->normal = outer_normal; out->material = s.material;
  return true;
}
bool prim_intersect(const Prim& p, const Ray& ray, Intersection* out) {          // triangle.rs:121-125
  return p.type == LR_PRIM_TRIANGLE ? triangle_intersect_mt(p, ray, out) : sphere_intersect(p, ray, out);
}
// SurfaceShape::sample  (triangle.rs:140-149, sphere.rs:79-84)
void prim_sample(const Prim& p, float xi_u, float xi_v, V3* value, float* pdf) {
  if (p.type == LR_PRIM_TRIANGLE) {
    float u = xi_u, v = xi_v;
    float mn = fmin_rs(u, v), mx = fmax_rs(u, v);
    *value = p.p0 * mn + p.p1 * (1.0f - mx) + p.p2 * (mx - mn);
    *pdf = 1.0f / p.area;
  } else {
    *value = p.centre + p.radius * sphere_uniform(xi_u, xi_v);
    *pdf = 1.0f / p.area;
  }
}
Prim make_prim(const LrPrimitive& lp) {
  Prim p; std::memset(&p, 0, sizeof(p));
  p.type = lp.type; p.material = lp.material;
  if (lp.type == LR_PRIM_TRIANGLE) {                                             // triangle.rs:25-40
    p.p0 = v3(lp.v[0], lp.v[1], lp.v[2]); p.p1 = v3(lp.v[3], lp.v[4], lp.v[5]); p.p2 = v3(lp.v[6], lp.v[7], lp.v[8]);
    p.normal = normalize(cross(p.p1 - p.p0, p.p2 - p.p0));
    p.area = norm(cross(p.p1 - p.p0, p.p2 - p.p0)) * 0.5f;
    p.bmin = v3(fmin_rs(fmin_rs(p.p0.x, p.p1.x), p.p2.x), fmin_rs(fmin_rs(p.p0.y, p.p1.y), p.p2.y), fmin_rs(fmin_rs(p.p0.z, p.p1.z), p.p2.z));
    p.bmax = v3(fmax_rs(fmax_rs(p.p0.x, p.p1.x), p.p2.x), fmax_rs(fmax_rs(p.p0.y, p.p1.y), p.p2.y), fmax_rs(fmax_rs(p.p0.z, p.p1.z), p.p2.z));
    p.bcentre = (p.bmax + p.bmin) / 2.0f;
  } else {                                                                       // sphere.rs:21-38
    p.centre = v3(lp.v[0], lp.v[1], lp.v[2]); p.radius = lp.v[3];
    p.area = 4.0f * PI * det_powi(p.radius, 2);
    V3 r = v3(p.radius, p.radius, p.radius);
    p.bmin = p.centre - r; p.bmax = p.centre + r; p.bcentre = p.centre;
  }
  return p;
}

// ------------------------------------------------------------------------------------------
// AABB + BVH  (aabb.rs, bvh.rs)
// ------------------------------------------------------------------------------------------
struct AABB { V3 mn, mx, centre; };
float aabb_surface_area(const AABB& a) {                                         // aabb.rs:17-28
  V3 side = v3(std::fabs(a.mx.x - a.mn.x), std::fabs(a.mx.y - a.mn.y), std::fabs(a.mx.z - a.mn.z));
  return 2.0f * (side.x * side.y + side.y * side.z + side.z * side.x);
}
AABB aabb_merge_with(const AABB& a, const AABB& b) {                             // aabb.rs:48-64
  AABB r;
  r.mn = v3(fmin_rs(a.mn.x, b.mn.x), fmin_rs(a.mn.y, b.mn.y), fmin_rs(a.mn.z, b.mn.z));
  r.mx = v3(fmax_rs(a.mx.x, b.mx.x), fmax_rs(a.mx.y, b.mx.y), fmax_rs(a.mx.z, b.mx.z));
  r.centre = (r.mn + r.mx) / 2.0f;
  return r;
}
bool aabb_is_intersect(const AABB& a, const Ray& ray, float pad) {               // aabb.rs:74-92
  float mn = -INF, mx = INF;
  for (int i = 0; i < 3; ++i) {
    float inv_d = 1.0f / comp(ray.direction, i);
    float t1 = ((comp(a.mn, i) - pad) - comp(ray.origin, i)) * inv_d;
    float t2 = ((comp(a.mx, i) + pad) - comp(ray.origin, i)) * inv_d;
    float t_min, t_max;
    if (t1 > t2) { t_min = t2; t_max = t1; } else { t_min = t1; t_max = t2; }
    if (mn < t_min) mn = t_min;
    if (mx > t_max) mx = t_max;
    if (mn > mx) return false;
  }
  return true;
}
struct BNode { AABB box; int left, right, leaf_index; };   // leaf_index >= 0 => Leaf (bvh.rs:9-13)
struct BVH {
  std::vector<BNode> nodes; int root;
  struct Leaf { AABB box; int index; };
  int construct(Leaf* list, int n) {                                             // bvh.rs:69-127
    const float t_aabb = 1.0f, t_tri = 2.0f;
    if (n == 1) { BNode b; b.box = list[0].box; b.left = b.right = -1; b.leaf_index = list[0].index; nodes.push_back(b); return (int)nodes.size() - 1; }
    AABB whole = list[0].box;
    int best_axis = 0, best_index = 0; float best_cost = 0.0f; bool have = false;
    for (int axis = 0; axis < 3; ++axis) {
      // the reference uses sort_unstable (tie order unspecified); a stable sort is one valid outcome
      std::stable_sort(list, list + n, [axis](const Leaf& a, const Leaf& b) { return comp(a.box.centre, axis) < comp(b.box.centre, axis); });
      std::vector<float> s1_a, s2_a; s1_a.reserve(n); s2_a.reserve(n);
      AABB s1 = list[0].box;
      for (int i = 0; i < n; ++i) { s1 = aabb_merge_with(s1, list[i].box); s1_a.push_back(aabb_surface_area(s1)); }
      AABB s2 = list[n - 1].box;
      for (int i = n - 1; i >= 1; --i) { s2 = aabb_merge_with(s2, list[i].box); s2_a.push_back(aabb_surface_area(s2)); }
      whole = aabb_merge_with(s1, list[n - 1].box);
      float s_a = aabb_surface_area(whole);
      int arg = 0; float cmin = 0.0f;
      for (int i = 0; i < n - 1; ++i) {
        float s1_n = (float)(i + 1), s2_n = (float)(n - i - 1);
        float c = 2.0f * t_aabb + (s1_a[i] * s1_n + s2_a[n - i - 2] * s2_n) * t_tri / s_a;
        if (!(c == c)) c = INFINITY;                                             // OrderedFloat puts NaN last (zero-area parents)
        if (i == 0 || c < cmin) { cmin = c; arg = i; }                           // min_by_key: first minimum
      }
      if (!have || cmin < best_cost) { have = true; best_cost = cmin; best_axis = axis; best_index = arg + 1; }
    }
    int axis = best_axis;
    std::stable_sort(list, list + n, [axis](const Leaf& a, const Leaf& b) { return comp(a.box.centre, axis) < comp(b.box.centre, axis); });
    int l = construct(list, best_index);
    int r = construct(list + best_index, n - best_index);
    BNode b; b.box = whole; b.left = l; b.right = r; b.leaf_index = -1; nodes.push_back(b);
    return (int)nodes.size() - 1;
  }
  void build(const std::vector<Prim>& prims) {                                   // bvh.rs:57-67
    std::vector<Leaf> leaf(prims.size());
    for (size_t i = 0; i < prims.size(); ++i) { leaf[i].box.mn = prims[i].bmin; leaf[i].box.mx = prims[i].bmax; leaf[i].box.centre = prims[i].bcentre; leaf[i].index = (int)i; }
    nodes.clear(); nodes.reserve(prims.size() * 2);
    root = prims.empty() ? -1 : construct(leaf.data(), (int)leaf.size());
  }
  // Slab test of aabb.rs:74-92 that also returns the clipped entry parameter (for mode 2's near-first order).
  static bool aabb_entry(const AABB& a, const Ray& ray, const V3& inv_d, float pad, float* t_entry) {
    float mn = -INF, mx = INF;
    for (int i = 0; i < 3; ++i) {
      float t1 = ((comp(a.mn, i) - pad) - comp(ray.origin, i)) * comp(inv_d, i);
      float t2 = ((comp(a.mx, i) + pad) - comp(ray.origin, i)) * comp(inv_d, i);
      float t_min, t_max;
      if (t1 > t2) { t_min = t2; t_max = t1; } else { t_min = t1; t_max = t2; }
      if (mn < t_min) mn = t_min;
      if (mx > t_max) mx = t_max;
      if (mn > mx) return false;
    }
    *t_entry = mn;
    return true;
  }
  void may_intersect(int ni, const Ray& ray, float pad, std::vector<int>& cand, uint64_t* visits) const {   // bvh.rs:20-25,38-45
    const BNode& b = nodes[ni];
    if (visits) ++*visits;
    if (!aabb_is_intersect(b.box, ray, pad)) return;
    if (b.leaf_index >= 0) { cand.push_back(b.leaf_index); return; }
    may_intersect(b.left, ray, pad, cand, visits);
    may_intersect(b.right, ray, pad, cand, visits);
  }
};

// ------------------------------------------------------------------------------------------
// cameras  (camera.rs) -- the constructors' outputs arrive in LrCamera
// ------------------------------------------------------------------------------------------
inline V3 arr3(const float* a) { return v3(a[0], a[1], a[2]); }
struct CamSample { Ray ray; float pdf; float g_term; };

CamSample camera_sample(const LrCamera& c, int x, int y, const Draw4& d) {
  CamSample r;
  V3 position = arr3(c.position), right = arr3(c.right), up = arr3(c.up), forward = arr3(c.forward);
  V3 aperture_position = arr3(c.aperture_position);
  if (c.type == LR_CAMERA_IDEAL_PINHOLE) {                                       // camera.rs:64-115
    float u = d.v[0], v = d.v[1];
    float px = ((((float)x + u) / (float)c.resolution[0]) - 0.5f) * c.sensor_size[0];
    float py = ((((float)y + v) / (float)c.resolution[1]) - 0.5f) * c.sensor_size[1];
    V3 point = position - right * px + up * py;
    float sensor_pdf = 1.0f, aperture_pdf = 1.0f;
    r.ray.origin = aperture_position;
    r.ray.direction = normalize(aperture_position - point);
    r.pdf = sensor_pdf * aperture_pdf;
    r.g_term = 1.0f;
  } else if (c.type == LR_CAMERA_THIN_LENS) {                                    // camera.rs:411-476
    float u = d.v[0], v = d.v[1];
    float px = ((((float)x + u) / (float)c.resolution[0]) - 0.5f) * c.sensor_size[0];
    float py = ((((float)y + v) / (float)c.resolution[1]) - 0.5f) * c.sensor_size[1];
    V3 point = position - right * px + up * py;
    float sensor_pdf = 1.0f / c.sensor_pixel_area;
    float au = 2.0f * PI * d.v[2];
    float av = std::sqrt(d.v[3]) * c.aperture_radius;
    float apx = det_cos(au) * av, apy = det_sin(au) * av;
    V3 apoint = aperture_position + right * apx + up * apy;
    float aperture_pdf = 1.0f / (PI * c.aperture_radius * c.aperture_radius);
    V3 sensor_center = aperture_position - point;
    V3 object_plane = sensor_center * (c.focus_distance / dot(sensor_center, forward));
    r.ray.origin = apoint;
    r.ray.direction = normalize(aperture_position + object_plane - apoint);
    r.pdf = sensor_pdf * aperture_pdf;
    V3 dir = normalize(apoint - point);                                          // geometry_term :436-455
    float cos_term = dot(dir, forward);
    float dd = c.aperture_sensor_distance / cos_term;
    r.g_term = cos_term * cos_term / (dd * dd);
  } else {                                                                       // camera.rs:168-188
    float u = d.v[0], v = d.v[1];
    float p = ((float)x + u) / (float)c.resolution[0] * PI * 2.0f;
    float t = ((float)y + v) / (float)c.resolution[1] * PI;
    r.ray.origin = aperture_position;
    r.ray.direction = v3(det_sin(t) * det_cos(p), det_sin(t) * det_sin(p), det_cos(t));
    r.pdf = 1.0f; r.g_term = 1.0f;
  }
  return r;
}

// ------------------------------------------------------------------------------------------
// Scene  (scene.rs, objects.rs, sky.rs)
// ------------------------------------------------------------------------------------------
struct Counters { uint64_t segments = 0, shadow_rays = 0, node_visits = 0, prim_tests = 0, sky_fetches = 0, samples = 0, tie_flips = 0, order_dependent = 0; };

struct Scene {
  std::vector<Prim> prims;
  std::vector<LrMaterial> materials;
  std::vector<int> emission;            // objects.rs:19-23: emitter list in instance order
  float emission_area = 0.0f;           // objects.rs:24
  BVH bvh;
  std::vector<int> rank;                // primitive -> its position in the reference tree's candidate order (bvh.rs:38-45: depth first, left first)
  LrSky sky;
  std::vector<float> texels;
  LrCamera camera;
  int depth, depth_limit; bool no_direct_emitter;
  int mode; float pad;

  V3 sky_radiance(const Ray& ray, Counters* ct) const {
    if (sky.type == LR_SKY_UNIFORM) return v3(sky.color[0], sky.color[1], sky.color[2]);   // sky.rs:17-21
    // IBLSky::radiance sky.rs:57-78
    if (ct) ct->sky_fetches++;
    float theta = det_acos(ray.direction.y);
    float phi = det_atan2(ray.direction.z, ray.direction.x);
    float uu = (phi + PI + sky.longitude_offset) / (2.0f * PI);
    float u = uu >= 0.0f ? det_fmod_pos(uu, 1.0f) : -det_fmod_pos(-uu, 1.0f);
    float vv = theta / PI;
    float v = vv >= 0.0f ? det_fmod_pos(vv, 1.0f) : -det_fmod_pos(-vv, 1.0f);
    size_t height = (size_t)sky.height, width = height * 2, all = width * height;
    float fx = std::floor((float)width * u), fy = std::floor((float)height * v);
    size_t x = (fx > 0.0f) ? (size_t)fx : 0, y = (fy > 0.0f) ? (size_t)fy : 0;   // `as usize` saturates (NaN, negatives -> 0)
    size_t index = (y * width + x) % all;
    return v3(texels[index * 3], texels[index * 3 + 1], texels[index * 3 + 2]);
  }

  // Leaf::may_intersect (bvh.rs:20-25) without the tree: aabb.rs:74-92 on the primitive's own exact box
  bool own_box_passes(int i, const Ray& ray) const {
    AABB own; own.mn = prims[(size_t)i].bmin; own.mx = prims[(size_t)i].bmax; own.centre = prims[(size_t)i].bcentre;
    return aabb_is_intersect(own, ray, 0.0f);
  }
  // Objects::intersect -> BVH::intersect (objects.rs:63, bvh.rs:131-141)
  bool intersect(const Ray& ray, Intersection* out, Counters* ct, bool shadow = false) const {
    bool found = false; Intersection best; best.distance = 0.0f; best.prim = -1;
    if (ct) { if (shadow) ct->shadow_rays++; else ct->segments++; }
    if (mode == 0) {
      for (size_t i = 0; i < prims.size(); ++i) {
        Intersection it;
        if (ct) ct->prim_tests++;
        if (prim_intersect(prims[i], ray, &it) && (!found || it.distance < best.distance)) { best = it; best.prim = (int)i; found = true; }
      }
    } else if (mode == 3 || mode == 7) {
      // bvh.rs:20-25 without the tree: the leaf's own box decides whether the primitive is a candidate; exact ties go to the
      // primitive that comes first in the reference's candidate order (mode 3) or to the lowest index (mode 7)
      for (size_t i = 0; i < prims.size(); ++i) {
        if (ct) ct->node_visits++;
        if (!own_box_passes((int)i, ray)) continue;
        Intersection it;
        if (ct) ct->prim_tests++;
        if (prim_intersect(prims[i], ray, &it) && (!found || it.distance < best.distance || (mode == 3 && it.distance == best.distance && rank[i] < rank[(size_t)best.prim]))) { best = it; best.prim = (int)i; found = true; }
      }
    } else if (mode == 2 || mode == 5) {
      // "optimized" CPU baseline: the same tree, walked near child first with an explicit
      // stack, subtrees entered beyond the best hit so far are skipped, no candidate vector.  Boxes only prune
      // (pad > 0 keeps them conservative), ties go to the lowest primitive index: same result as mode 0.
      int stack[256]; float stack_t[256]; int sp = 0;
      bool overflow = false;
      V3 inv_d = v3(1.0f / ray.direction.x, 1.0f / ray.direction.y, 1.0f / ray.direction.z);
      float te;
      if (bvh.root >= 0 && BVH::aabb_entry(bvh.nodes[bvh.root].box, ray, inv_d, pad, &te)) { stack[sp] = bvh.root; stack_t[sp++] = te; }
      if (ct && bvh.root >= 0) ct->node_visits++;
      while (sp > 0) {
        --sp;
        int ni = stack[sp];
        if (found && stack_t[sp] > best.distance + pad) continue;
        const BNode& b = bvh.nodes[ni];
        if (b.leaf_index >= 0) {
          Intersection it; int i = b.leaf_index;
          if (ct) ct->prim_tests++;
          if (prim_intersect(prims[i], ray, &it) && (mode != 5 || own_box_passes(i, ray))) {
            bool better = !found || it.distance < best.distance || (it.distance == best.distance && (mode == 5 ? rank[(size_t)i] < rank[(size_t)best.prim] : i < best.prim));
            if (better) { best = it; best.prim = i; found = true; }
          }
          continue;
        }
        float tl = 0.0f, tr = 0.0f;
        bool hl = BVH::aabb_entry(bvh.nodes[b.left].box, ray, inv_d, pad, &tl);
        bool hr = BVH::aabb_entry(bvh.nodes[b.right].box, ray, inv_d, pad, &tr);
        if (ct) ct->node_visits += 2;
        if (hl && hr) {
          if (tl <= tr) { stack[sp] = b.right; stack_t[sp++] = tr; stack[sp] = b.left; stack_t[sp++] = tl; }
          else { stack[sp] = b.left; stack_t[sp++] = tl; stack[sp] = b.right; stack_t[sp++] = tr; }
        } else if (hl) { stack[sp] = b.left; stack_t[sp++] = tl; }
        else if (hr) { stack[sp] = b.right; stack_t[sp++] = tr; }
        if (sp > 250) { overflow = true; break; }                                 // a pathologically deep tree: redo this ray the literal way below
      }
      if (overflow) {
        found = false; best.distance = 0.0f; best.prim = -1;
        std::vector<int> cand;
        bvh.may_intersect(bvh.root, ray, pad, cand, nullptr);
        for (size_t k = 0; k < cand.size(); ++k) {
          Intersection it; int i = cand[k];
          if (prim_intersect(prims[i], ray, &it) && (mode != 5 || own_box_passes(i, ray)) &&
              (!found || it.distance < best.distance || (it.distance == best.distance && (mode == 5 ? rank[(size_t)i] < rank[(size_t)best.prim] : i < best.prim)))) { best = it; best.prim = i; found = true; }
        }
      }
    } else {
      std::vector<int> cand;
      if (bvh.root >= 0) bvh.may_intersect(bvh.root, ray, (mode == 4 || mode == 6 || mode == 8) ? 0.0f : pad, cand, ct ? &ct->node_visits : nullptr);
      for (size_t k = 0; k < cand.size(); ++k) {
        Intersection it; int i = cand[k];
        if (ct) ct->prim_tests++;
        if (prim_intersect(prims[i], ray, &it)) {
          // min_by keeps the first minimum in candidate order; with pad > 0 (conservative mode)
          // ties resolve to the lowest primitive index so that the result equals mode 0
          // (mode 8: ties to the lowest index, the rule of mode 7)
          bool better = !found || it.distance < best.distance || ((pad > 0.0f || mode == 8) && it.distance == best.distance && i < best.prim);
          if (better) { best = it; best.prim = i; found = true; }
        }
      }
      if (mode == 6 && ct) {
        // AUDIT of the literal walk (mode 1 at pad 0 is what was just evaluated): the definition (mode 3: every primitive behind its
        // own box, lowest index on ties) on the same ray.  tie_flips = same distance bits, another primitive (an exact tie the
        // reference's candidate order decides); order_dependent = anything else -- a query whose result would depend on the tree.
        // (scenes of more than 64 primitives: the definition's tie rule on the walk's own candidate list -- that the list holds
        //  every primitive whose own box passes is checked on ray batches, tests/test_oracle_properties.py)
        bool f3 = false; Intersection b3; b3.distance = 0.0f; b3.prim = -1;
        if (prims.size() <= 64) {
          for (size_t i = 0; i < prims.size(); ++i) {
            if (!own_box_passes((int)i, ray)) continue;
            Intersection it;
            if (prim_intersect(prims[i], ray, &it) && (!f3 || it.distance < b3.distance || (it.distance == b3.distance && rank[i] < rank[(size_t)b3.prim]))) { b3 = it; b3.prim = (int)i; f3 = true; }
          }
        } else {
          for (size_t k = 0; k < cand.size(); ++k) {
            Intersection it; int i = cand[k];
            if (prim_intersect(prims[(size_t)i], ray, &it) && (!f3 || it.distance < b3.distance || (it.distance == b3.distance && rank[(size_t)i] < rank[(size_t)b3.prim]))) { b3 = it; b3.prim = i; f3 = true; }
          }
        }
        if (f3 != found || (found && b3.distance != best.distance)) ct->order_dependent++;
        else if (found && b3.prim != best.prim) ct->tie_flips++;
      }
    }
    if (found) *out = best;
    return found;
  }

  float russian_roulette(float init, int d) const {                              // scene.rs:64-76
    float p = init;
    if (d > depth_limit) p *= det_powi(0.5f, d - depth_limit);
    if (d <= depth && p > 0.0f) p = 1.0f;
    return p;
  }

  // Objects::sample_emission objects.rs:37-51
  void sample_emission(const Draw4& d, V3* value, float* pdf) const {
    float roulette = emission_area * d.v[1];
    float area = 0.0f;
    for (size_t k = 0; k < emission.size(); ++k) {
      const Prim& obj = prims[emission[k]];
      area += obj.area;
      if (roulette <= area) {
        V3 val; float spdf;
        prim_sample(obj, d.v[2], d.v[3], &val, &spdf);
        *value = val; *pdf = spdf * obj.area / emission_area;
        return;
      }
    }
    // unreachable!() in the reference (only through float round-off of the running sum)
    const Prim& obj = prims[emission.back()];
    V3 val; float spdf; prim_sample(obj, d.v[2], d.v[3], &val, &spdf);
    *value = val; *pdf = spdf * obj.area / emission_area;
  }

  V3 direct_light_radiance(const Intersection& i, const Ray& ray, const Rng& rng, int d, Counters* ct) const {   // scene.rs:104-151
    const LrMaterial& m = materials[i.material];
    if (sqr_norm(material_emission(m)) > 0.0f || !(emission_area > 0.0f)) return v3(0, 0, 0);
    Draw4 dr = rng.block(1 + 2 * (uint32_t)d);
    V3 sample_value; float sample_pdf;
    sample_emission(dr, &sample_value, &sample_pdf);
    V3 direct_path = sample_value - i.position;
    Ray direct_ray; direct_ray.origin = i.position; direct_ray.direction = normalize(direct_path);
    V3 point_in = direct_ray.direction;
    V3 point_out = -ray.direction;
    V3 point_normal = orienting_normal(point_out, i.normal);
    if (dot(point_in, point_normal) <= 0.0f) return v3(0, 0, 0);
    Intersection direct_i;
    if (!intersect(direct_ray, &direct_i, ct, true)) return v3(0, 0, 0);
    if (std::fabs(direct_i.distance - norm(direct_path)) > EPS) return v3(0, 0, 0);
    V3 light_out = -direct_ray.direction;
    V3 light_normal = direct_i.normal;
    float light_cos = dot(light_out, light_normal);
    if (light_cos <= 0.0f) return v3(0, 0, 0);
    float point_cos = dot(point_in, point_normal);
    float g_term = point_cos * light_cos / sqr_norm(direct_path);
    V3 brdf = material_brdf(m, point_out, point_in, point_normal, i.position);
    V3 l_i = material_emission(materials[direct_i.material]);
    float pdf = sample_pdf;
    return brdf * l_i * g_term / pdf;
  }

  // scene.rs:78-102 with the recursive call made explicit
  V3 material_interaction(const Intersection& i, const Ray& ray, const Rng& rng, int d, bool nee, Counters* ct) const {
    const LrMaterial& m = materials[i.material];
    V3 out_ = -ray.direction;
    Draw4 dr = rng.block(2 + 2 * (uint32_t)d);
    V3 in_; float pdf;
    material_sample(m, out_, i.normal, dr.v, &in_, &pdf);
    V3 brdf = material_brdf(m, out_, in_, i.normal, i.position);
    V3 coef = material_coef(m, out_, i.normal, i.distance);
    float c = dot(in_, i.normal);
    Ray new_ray; new_ray.direction = in_; new_ray.origin = i.position;
    V3 l_i = nee ? radiance_nee_recursive(new_ray, rng, d + 1, true, ct) : radiance_recursive(new_ray, rng, d + 1, ct);
    return brdf * coef * l_i * c / pdf;
  }

  V3 radiance_recursive(const Ray& ray, const Rng& rng, int d, Counters* ct) const {        // scene.rs:24-32
    Intersection i;
    if (!intersect(ray, &i, ct)) return sky_radiance(ray, ct);
    // intersect_radiance scene.rs:153-171
    const LrMaterial& m = materials[i.material];
    V3 l_e = (!(no_direct_emitter && d == 0) && dot(-ray.direction, i.normal) > 0.0f) ? material_emission(m) : v3(0, 0, 0);
    float p = russian_roulette(material_weight(m), d);
    if (p != 1.0f && rng.block(1 + 2 * (uint32_t)d).v[0] >= p) return l_e;
    V3 material_radiance = material_interaction(i, ray, rng, d, false, ct);
    return l_e + material_radiance / p;
  }
  V3 radiance_nee_recursive(const Ray& ray, const Rng& rng, int d, bool no_emission, Counters* ct) const {   // scene.rs:38-46
    Intersection i;
    if (!intersect(ray, &i, ct)) return sky_radiance(ray, ct);
    // intersect_radiance_nee scene.rs:173-193
    const LrMaterial& m = materials[i.material];
    V3 l_e = (!(no_direct_emitter && d == 0) && !no_emission && dot(-ray.direction, i.normal) > 0.0f) ? material_emission(m) : v3(0, 0, 0);
    float p = russian_roulette(material_weight(m), d);
    if (p != 1.0f && rng.block(1 + 2 * (uint32_t)d).v[0] >= p) return l_e;
    V3 direct = direct_light_radiance(i, ray, rng, d, ct);
    V3 material_radiance = material_interaction(i, ray, rng, d, true, ct);
    return l_e + (direct + material_radiance) / p;
  }
};

bool build_scene(const LrSceneDesc* desc, const LrRenderParams* params, int mode, float pad, Scene* s) {
  if (!desc || desc->abi_version != LR_ABI_VERSION) return false;
  s->materials.assign(desc->materials, desc->materials + desc->n_materials);
  s->prims.clear(); s->prims.reserve(desc->n_prims);
  for (int i = 0; i < desc->n_prims; ++i) {
    if (desc->prims[i].material < 0 || desc->prims[i].material >= desc->n_materials) return false;
    s->prims.push_back(make_prim(desc->prims[i]));
  }
  s->emission.clear(); s->emission_area = 0.0f;
  for (size_t i = 0; i < s->prims.size(); ++i)                                   // objects.rs:19-24
    if (sqr_norm(material_emission(s->materials[s->prims[i].material])) > 0.0f) s->emission.push_back((int)i);
  for (size_t k = 0; k < s->emission.size(); ++k) s->emission_area += s->prims[s->emission[k]].area;
  s->sky = desc->sky;
  if (desc->sky.type == LR_SKY_IBL) {
    size_t n = (size_t)desc->sky.height * (size_t)desc->sky.height * 2 * 3;
    s->texels.assign(desc->sky.texels, desc->sky.texels + n);
  }
  s->camera = desc->camera;
  s->depth = params ? params->depth : 5; s->depth_limit = params ? params->depth_limit : 64;
  s->no_direct_emitter = params ? params->no_direct_emitter != 0 : false;
  s->mode = mode; s->pad = pad;
  if (mode != 0 && mode != 7) s->bvh.build(s->prims);
  s->rank.assign(s->prims.size(), 0);
  if (mode != 0 && mode != 7 && s->bvh.root >= 0) {
    std::vector<int> stack(1, s->bvh.root); int k = 0;
    while (!stack.empty()) {
      const BNode& b = s->bvh.nodes[(size_t)stack.back()]; stack.pop_back();
      if (b.leaf_index >= 0) { s->rank[(size_t)b.leaf_index] = k++; continue; }
      stack.push_back(b.right); stack.push_back(b.left);
    }
  }
  return true;
}

// main.rs:92-121: one pixel = fold over spp samples, then / spp
V3 render_pixel(const Scene& s, const LrRenderParams& p, int x, int y, Counters* ct) {
  V3 sum = v3(0, 0, 0);
  uint32_t pixel = (uint32_t)y * (uint32_t)s.camera.resolution[0] + (uint32_t)x;
  float sens = s.camera.type == LR_CAMERA_THIN_LENS ? s.camera.sensor_sensitivity : 1.0f;   // camera.rs:117-119,330-332
  for (int k = 0; k < p.spp; ++k) {
    Rng rng = {p.seed, pixel, (uint32_t)k};
    CamSample cs = camera_sample(s.camera, x, y, rng.block(0));
    V3 l = p.integrator == LR_INTEGRATOR_PT ? s.radiance_recursive(cs.ray, rng, 0, ct)
                                            : s.radiance_nee_recursive(cs.ray, rng, 0, false, ct);
    V3 e = l * cs.g_term;
    V3 delta = e * (sens / cs.pdf);
    sum = sum + delta;
    if (ct) ct->samples++;
  }
  return sum / (float)p.spp;
}

}  // namespace

// ==========================================================================================
// C entry points (ctypes)
// ==========================================================================================
extern "C" {

struct LrOracleStats {
  uint64_t samples, segments, shadow_rays, node_visits, prim_tests, sky_fetches;
  double seconds;
  uint64_t tie_flips, order_dependent;     // mode 6 (audit of the literal walk against the definition, see Scene::intersect)
};

// mode 0: brute force over all primitives; mode 1: reference SAH tree + collect-all traversal, boxes padded by `pad`
// (pad = 0 is the literal reference); mode 2: the same tree walked near-first with early-out (the "optimized"
// CPU baseline; needs pad > 0 to stay exact); mode 3: every primitive behind its own exact box (the definition the
// HIP path implements); mode 4: mode 3 through the reference's tree (pad is ignored); mode 5: mode 3 through the near-first walk of mode 2 (pad > 0).  n_threads <= 0 -> hardware_concurrency.
int lr_oracle_render(const LrSceneDesc* desc, const LrRenderParams* params, const LrTile* tiles, int n_tiles,
                     float* rgb_out, size_t row_stride_floats, int n_threads, int mode, float pad, LrOracleStats* stats) {
  if (!desc || !params || (!tiles && n_tiles > 0) || !rgb_out || params->spp <= 0) return LR_EINVAL;
  Scene s;
  if (!build_scene(desc, params, mode, pad, &s)) return LR_EINVAL;
  int W = s.camera.resolution[0], H = s.camera.resolution[1];
  // work list: (tile, row) pairs; main.rs:73-126 farms single pixels, rows only change scheduling
  struct Row { int x0, w, y; };
  std::vector<Row> rows;
  for (int t = 0; t < n_tiles; ++t) {
    const LrTile& tl = tiles[t];
    if (tl.w < 0 || tl.h < 0 || tl.x0 < 0 || tl.y0 < 0 || tl.x0 + tl.w > W || tl.y0 + tl.h > H) return LR_EINVAL;
    for (int y = tl.y0; y < tl.y0 + tl.h; ++y) if (tl.w > 0) rows.push_back({tl.x0, tl.w, y});
  }
  if (n_threads <= 0) n_threads = (int)std::thread::hardware_concurrency();
  if (n_threads <= 0) n_threads = 1;
  std::atomic<size_t> next(0);
  std::vector<Counters> counters((size_t)n_threads);
  auto t0 = std::chrono::steady_clock::now();
  auto worker = [&](int tid) {
    Counters local;                                   // thread-private (no false sharing); published once at the end
    Counters* ct = stats ? &local : nullptr;
    for (;;) {
      size_t r = next.fetch_add(1);
      if (r >= rows.size()) break;
      const Row& row = rows[r];
      for (int x = row.x0; x < row.x0 + row.w; ++x) {
        V3 px = render_pixel(s, *params, x, row.y, ct);
        float* o = rgb_out + (size_t)row.y * row_stride_floats + (size_t)x * 3;
        o[0] = px.x; o[1] = px.y; o[2] = px.z;
      }
    }
    counters[(size_t)tid] = local;
  };
  std::vector<std::thread> th;
  for (int i = 1; i < n_threads; ++i) th.emplace_back(worker, i);
  worker(0);
  for (auto& t : th) t.join();
  auto t1 = std::chrono::steady_clock::now();
  if (stats) {
    std::memset(stats, 0, sizeof(*stats));
    for (auto& c : counters) {
      stats->samples += c.samples; stats->segments += c.segments; stats->shadow_rays += c.shadow_rays;
      stats->node_visits += c.node_visits; stats->prim_tests += c.prim_tests; stats->sky_fetches += c.sky_fetches;
      stats->tie_flips += c.tie_flips; stats->order_dependent += c.order_dependent;
    }
    stats->seconds = std::chrono::duration<double>(t1 - t0).count();
  }
  return LR_OK;
}

// The candidate order of the reference's tree (bvh.rs:38-45: left subtree first): order_out[k] = primitive index of the k-th leaf
// of a depth-first walk.  bvh.rs:131-141 keeps the first minimum in this order; the host builder's bvh_prim_order must equal it.
int lr_oracle_bvh_leaf_order(const LrSceneDesc* desc, int32_t* order_out) {
  Scene s; if (!build_scene(desc, nullptr, 1, 0.0f, &s)) return LR_EINVAL;
  std::vector<int> stack; int k = 0;
  if (s.bvh.root >= 0) stack.push_back(s.bvh.root);
  while (!stack.empty()) {
    const BNode& b = s.bvh.nodes[(size_t)stack.back()]; stack.pop_back();
    if (b.leaf_index >= 0) { order_out[k++] = b.leaf_index; continue; }
    stack.push_back(b.right); stack.push_back(b.left);
  }
  return k;
}

// ---- unit hooks for known-answer tests ------------------------------------------------------
// variant 0 = intersect_mt, 1 = intersect_3c; out = distance, position[3], normal[3]
int lr_oracle_triangle_intersect(const float* p9, const float* o, const float* d, int variant, float* out7) {
  LrPrimitive lp; std::memset(&lp, 0, sizeof(lp)); lp.type = LR_PRIM_TRIANGLE; std::memcpy(lp.v, p9, 36);
  Prim p = make_prim(lp);
  Ray r; r.origin = arr3(o); r.direction = arr3(d);
  Intersection it;
  bool hit = variant == 0 ? triangle_intersect_mt(p, r, &it) : triangle_intersect_3c(p, r, &it);
  if (!hit) return 0;
  out7[0] = it.distance; out7[1] = it.position.x; out7[2] = it.position.y; out7[3] = it.position.z;
  out7[4] = it.normal.x; out7[5] = it.normal.y; out7[6] = it.normal.z;
  return 1;
}
int lr_oracle_sphere_intersect(const float* c, float radius, const float* o, const float* d, float* out7) {
  LrPrimitive lp; std::memset(&lp, 0, sizeof(lp)); lp.type = LR_PRIM_SPHERE; lp.v[0] = c[0]; lp.v[1] = c[1]; lp.v[2] = c[2]; lp.v[3] = radius;
  Prim p = make_prim(lp);
  Ray r; r.origin = arr3(o); r.direction = arr3(d);
  Intersection it;
  if (!sphere_intersect(p, r, &it)) return 0;
  out7[0] = it.distance; out7[1] = it.position.x; out7[2] = it.position.y; out7[3] = it.position.z;
  out7[4] = it.normal.x; out7[5] = it.normal.y; out7[6] = it.normal.z;
  return 1;
}
void lr_oracle_reflect(const float* v, const float* n, float* out3) {
  V3 r = reflect(arr3(v), arr3(n)); out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}
int lr_oracle_refract(const float* v, const float* n, float ratio, float* out3) {
  V3 r; if (!refract(arr3(v), arr3(n), ratio, &r)) return 0;
  out3[0] = r.x; out3[1] = r.y; out3[2] = r.z; return 1;
}
void lr_oracle_orthonormal_basis(const float* w, float* t3, float* b3) {
  V3 t, b; orthonormal_basis(arr3(w), &t, &b);
  t3[0] = t.x; t3[1] = t.y; t3[2] = t.z; b3[0] = b.x; b3[1] = b.y; b3[2] = b.z;
}
void lr_oracle_material_brdf(const LrMaterial* m, const float* out_, const float* in_, const float* n, const float* pos, float* rgb) {
  V3 r = material_brdf(*m, arr3(out_), arr3(in_), arr3(n), arr3(pos)); rgb[0] = r.x; rgb[1] = r.y; rgb[2] = r.z;
}
void lr_oracle_material_sample(const LrMaterial* m, const float* out_, const float* n, const float* xi3, float* in3, float* pdf) {
  V3 i; material_sample(*m, arr3(out_), arr3(n), xi3, &i, pdf); in3[0] = i.x; in3[1] = i.y; in3[2] = i.z;
}
float lr_oracle_material_weight(const LrMaterial* m) { return material_weight(*m); }
void lr_oracle_material_coef(const LrMaterial* m, const float* out_, const float* n, float dist, float* rgb) {
  V3 r = material_coef(*m, arr3(out_), arr3(n), dist); rgb[0] = r.x; rgb[1] = r.y; rgb[2] = r.z;
}
void lr_oracle_ior_pair(const LrMaterial* m, const float* out_, const float* n, float* pair2) { ior_pair(*m, arr3(out_), arr3(n), &pair2[0], &pair2[1]); }
float lr_oracle_fresnel(float from_ior, float to_ior, const float* out_, const float* in_, const float* on) {
  return fresnel_exact(from_ior, to_ior, arr3(out_), arr3(in_), arr3(on));
}
void lr_oracle_checker(float u, float v, float* rgb) { V3 r = checker(u, v); rgb[0] = r.x; rgb[1] = r.y; rgb[2] = r.z; }
float lr_oracle_russian_roulette(float init, int d, int depth, int depth_limit) {
  Scene s; s.depth = depth; s.depth_limit = depth_limit; return s.russian_roulette(init, d);
}
void lr_oracle_rng_block(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t block, float* out4) {
  Draw4 d = rng_block(seed, pixel, sample, block); std::memcpy(out4, d.v, 16);
}
void lr_oracle_camera_sample(const LrCamera* c, int x, int y, const float* xi4, float* out8) {
  Draw4 d; std::memcpy(d.v, xi4, 16);
  CamSample s = camera_sample(*c, x, y, d);
  out8[0] = s.ray.origin.x; out8[1] = s.ray.origin.y; out8[2] = s.ray.origin.z;
  out8[3] = s.ray.direction.x; out8[4] = s.ray.direction.y; out8[5] = s.ray.direction.z;
  out8[6] = s.pdf; out8[7] = s.g_term;
}
void lr_oracle_prim_sample(const LrPrimitive* lp, float u, float v, float* out4) {
  Prim p = make_prim(*lp); V3 val; float pdf; prim_sample(p, u, v, &val, &pdf);
  out4[0] = val.x; out4[1] = val.y; out4[2] = val.z; out4[3] = pdf;
}
// closest hit over a batch of rays: out[i] = (prim index or -1, distance bits)
int lr_oracle_intersect_batch(const LrSceneDesc* desc, int mode, float pad, int n, const float* origins, const float* dirs, int32_t* prim_out, float* t_out) {
  Scene s; if (!build_scene(desc, nullptr, mode, pad, &s)) return LR_EINVAL;
  int n_threads = (int)std::thread::hardware_concurrency();
  if (n_threads <= 0) n_threads = 1;
  if (n < 4096) n_threads = 1;
  std::atomic<int> next(0);
  auto worker = [&]() {
    for (;;) {
      int b = next.fetch_add(256);
      if (b >= n) break;
      int e = b + 256 < n ? b + 256 : n;
      for (int i = b; i < e; ++i) {
        Ray r; r.origin = arr3(origins + 3 * (size_t)i); r.direction = arr3(dirs + 3 * (size_t)i);
        Intersection it;
        if (s.intersect(r, &it, nullptr)) { prim_out[i] = it.prim; t_out[i] = it.distance; }
        else { prim_out[i] = -1; t_out[i] = 0.0f; }
      }
    }
  };
  std::vector<std::thread> th;
  for (int i = 1; i < n_threads; ++i) th.emplace_back(worker);
  worker();
  for (auto& t : th) t.join();
  return LR_OK;
}
// the deterministic math spec over arrays (fn numbering = the device's lr_selftest_math): 0 sin 1 cos 2 acos 3 atan2 4 pow 5 exp 6 fmod_pos
int lr_oracle_math_batch(int fn, const float* a, const float* b, float* out, int n) {
  for (int i = 0; i < n; ++i) {
    float x = a[i], y = b ? b[i] : 0.0f, r = 0.0f;
    switch (fn) {
      case 0: r = det_sin(x); break;
      case 1: r = det_cos(x); break;
      case 2: r = det_acos(x); break;
      case 3: r = det_atan2(x, y); break;
      case 4: r = det_pow(x, y); break;
      case 5: r = det_exp(x); break;
      case 6: r = det_fmod_pos(x, y); break;
      default: return LR_EINVAL;
    }
    out[i] = r;
  }
  return LR_OK;
}
// Objects::sample_emission's pick (objects.rs:37-51): k_out[i] = index into the emitter list for roulette = area * xi[i]
int lr_oracle_emitter_pick(const LrSceneDesc* desc, int n, const float* xi, int32_t* k_out) {
  Scene s; if (!build_scene(desc, nullptr, 0, 0.0f, &s)) return LR_EINVAL;
  if (s.emission.empty()) return LR_EINVAL;
  for (int i = 0; i < n; ++i) {
    float roulette = s.emission_area * xi[i];
    float area = 0.0f; int pick = (int)s.emission.size() - 1;     // the reference's unreachable!() tail -> last emitter
    for (size_t k = 0; k < s.emission.size(); ++k) {
      area += s.prims[s.emission[k]].area;
      if (roulette <= area) { pick = (int)k; break; }
    }
    k_out[i] = pick;
  }
  return (int)s.emission.size();
}
// Objects::sample_emission in full (objects.rs:37-51 + triangle.rs:140-149 / sphere.rs:79-84): out4[4*i..] = point.xyz, pdf
int lr_oracle_emission_sample(const LrSceneDesc* desc, int n, const float* xi4, float* out4) {
  Scene s; if (!build_scene(desc, nullptr, 0, 0.0f, &s)) return LR_EINVAL;
  if (s.emission.empty()) return LR_EINVAL;
  for (int i = 0; i < n; ++i) {
    Draw4 d; std::memcpy(d.v, xi4 + 4 * (size_t)i, 16);
    V3 val; float pdf; s.sample_emission(d, &val, &pdf);
    out4[4 * (size_t)i] = val.x; out4[4 * (size_t)i + 1] = val.y; out4[4 * (size_t)i + 2] = val.z; out4[4 * (size_t)i + 3] = pdf;
  }
  return 0;
}
// AABB::is_intersect (aabb.rs:74-92), literal: box6 = min.xyz, max.xyz
int lr_oracle_aabb_is_intersect(const float* box6, const float* o, const float* d) {
  AABB a; a.mn = arr3(box6); a.mx = arr3(box6 + 3);
  Ray r; r.origin = arr3(o); r.direction = arr3(d);
  return aabb_is_intersect(a, r, 0.0f) ? 1 : 0;
}
void lr_oracle_sky_radiance(const LrSceneDesc* desc, const float* dir, float* rgb) {
  Scene s; if (!build_scene(desc, nullptr, 0, 0.0f, &s)) { rgb[0] = rgb[1] = rgb[2] = 0; return; }
  Ray r; r.origin = v3(0, 0, 0); r.direction = arr3(dir);
  V3 c = s.sky_radiance(r, nullptr); rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
}
void lr_oracle_sky_batch(const LrSceneDesc* desc, int n, const float* dirs, float* rgb) {
  Scene s; if (!build_scene(desc, nullptr, 0, 0.0f, &s)) return;
  for (int i = 0; i < n; ++i) {
    Ray r; r.origin = v3(0, 0, 0); r.direction = arr3(dirs + 3 * (size_t)i);
    V3 c = s.sky_radiance(r, nullptr); rgb[3 * (size_t)i] = c.x; rgb[3 * (size_t)i + 1] = c.y; rgb[3 * (size_t)i + 2] = c.z;
  }
}
// math (compared with numpy in tests)
float lr_oracle_sin(float x) { return det_sin(x); }
float lr_oracle_cos(float x) { return det_cos(x); }
float lr_oracle_acos(float x) { return det_acos(x); }
float lr_oracle_atan2(float y, float x) { return det_atan2(y, x); }
float lr_oracle_pow(float x, float y) { return det_pow(x, y); }
float lr_oracle_exp(float x) { return det_exp(x); }
float lr_oracle_fmod_pos(float x, float k) { return det_fmod_pos(x, k); }

}  // extern "C"
