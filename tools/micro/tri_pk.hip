// ALU cost of triangle.rs:69-100 for TWO triangles with packed f32 arithmetic (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32, tri_test2
// below) against two scalar tests (lr_path.h tri_test_bf), data in registers, 8 waves per SIMD.  Round 4 result on MI355X: 2.895 ms
// packed vs 2.623 ms scalar -- a packed instruction costs ~2.1 plain issue slots in this mix (38 packed + 12 plain against 84 plain),
// so packing buys nothing; tools/micro/pk_rate.hip / pk_port.hip had suggested 1.1x, but their plain baseline ran at half rate
// (v_fma_f32 vN, vN, v1, v9: both constant operands in VGPR bank 1).  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -I lumillyrender_amd/csrc -o build/tri_pk tools/micro/tri_pk.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include "lr_math.h"
namespace lr {
typedef float f2 __attribute__((ext_vector_type(2)));
struct P3 { f2 x, y, z; };                                  // two 3-vectors side by side: .x = {a.x, b.x} ...

LR_DEV f2 sp2(float a) { return (f2){a, a}; }               // both halves the same value (op_sel picks the register: no move)
LR_DEV f2 mk2(float a, float b) { return (f2){a, b}; }
LR_DEV P3 p3(f2 x, f2 y, f2 z) { P3 r; r.x = x; r.y = y; r.z = z; return r; }
LR_DEV P3 splat3(V3 a) { return p3(sp2(a.x), sp2(a.y), sp2(a.z)); }
LR_DEV P3 pair3(V3 a, V3 b) { return p3(mk2(a.x, b.x), mk2(a.y, b.y), mk2(a.z, b.z)); }
LR_DEV P3 operator-(P3 a, P3 b) { return p3(a.x - b.x, a.y - b.y, a.z - b.z); }
LR_DEV f2 dot2(P3 a, P3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }                       // vector3.rs:77-81, twice
LR_DEV P3 cross2(P3 a, P3 b) {                                                                  // vector3.rs:83-91, twice
  return p3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
LR_DEV f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
// rcp_exact_mid of lr_math.h for both halves: the same v_rcp_f32 seed and the same two fused Newton steps
LR_DEV f2 rcp_exact_mid2(f2 d) {
  f2 r = mk2(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y));
  f2 e = fma2(-d, r, sp2(1.0f));
  r = fma2(e, r, r);
  e = fma2(-d, r, sp2(1.0f));
  return fma2(e, r, r);
}

// triangle.rs:69-100 (tri_test_bf's operations) for two triangles {A, B} against two rays {A, B}: pass a splat for whatever the
// two halves share.  Outputs per half: t and the acceptance of triangle.rs:75,80,85,90.
struct Hit2 { float tA, tB; bool hA, hB; };
LR_DEV Hit2 tri_test2(P3 p0, P3 e1, P3 e2, P3 o, P3 d) {
  P3 pv = cross2(d, e2);
  f2 det = dot2(e1, pv);
  f2 invdet = rcp_exact_mid2(det);
  P3 tv = o - p0;
  f2 u = dot2(tv, pv) * invdet;
  P3 qv = cross2(tv, e1);
  f2 v = dot2(d, qv) * invdet;
  f2 t = dot2(e2, qv) * invdet;
  f2 uv = u + v;
  Hit2 h;
  h.tA = t.x; h.tB = t.y;
  h.hA = bool(!(__builtin_fabsf(det.x) < kEps)) & bool(!(u.x < 0.0f)) & bool(!(u.x > 1.0f)) & bool(!(v.x < 0.0f)) & bool(!(uv.x > 1.0f)) & bool(!(t.x < kEps));
  h.hB = bool(!(__builtin_fabsf(det.y) < kEps)) & bool(!(u.y < 0.0f)) & bool(!(u.y > 1.0f)) & bool(!(v.y < 0.0f)) & bool(!(uv.y > 1.0f)) & bool(!(t.y < kEps));
  return h;
}

}
using namespace lr;
LR_DEV bool tri_test_bf(V3 p0, V3 e1, V3 e2, V3 o, V3 d, float* t_out) {
  V3 pv = cross(d, e2);
  float det = dot(e1, pv);
  float invdet = rcp_exact_mid(det);
  V3 tv = o - p0;
  float u = dot(tv, pv) * invdet;
  V3 qv = cross(tv, e1);
  float v = dot(d, qv) * invdet;
  float t = dot(e2, qv) * invdet;
  *t_out = t;
  return bool(!(__builtin_fabsf(det) < kEps)) & bool(!(u < 0.0f)) & bool(!(u > 1.0f)) & bool(!(v < 0.0f)) & bool(!(u + v > 1.0f)) & bool(!(t < kEps));
}
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, float seed, int iters) {
  V3 pa = v3(seed + threadIdx.x, 1, 2), e1a = v3(1, seed, 0.5f), e2a = v3(0.25f, 1, seed);
  V3 pb = v3(seed - threadIdx.x, 2, 1), e1b = v3(2, seed, 0.75f), e2b = v3(0.5f, 2, seed);
  V3 o = v3(0.1f * threadIdx.x, 0.2f, 0.3f), d = v3(0.3f, 0.5f, 0.8f);
  float best = 3e38f;
  for (int it = 0; it < iters; ++it) {
    float tA, tB; bool hA, hB;
    if (MODE == 0) { hA = tri_test_bf(pa, e1a, e2a, o, d, &tA); hB = tri_test_bf(pb, e1b, e2b, o, d, &tB); }
    else { Hit2 h = tri_test2(pair3(pa, pb), pair3(e1a, e1b), pair3(e2a, e2b), splat3(o), splat3(d)); tA = h.tA; tB = h.tB; hA = h.hA; hB = h.hB; }
    if (hA && tA < best) best = tA;
    if (hB && tB < best) best = tB;
    o.x += 1e-3f; asm volatile("" : "+v"(pa.x), "+v"(pb.x), "+v"(o.y));
  }
  out[blockIdx.x * 256 + threadIdx.x] = best;
}
template <int MODE> double run(float* d, int iters) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(256 * 8), dim3(256), 0, 0, d, 1.5f, 10);
  (void)hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(256 * 8), dim3(256), 0, 0, d, 1.5f, iters);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  return ms;
}
int main() {
  float* d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
  const int iters = 4000;
  double s = run<0>(d, iters), p = run<1>(d, iters);
  printf("two scalar triangle tests: %.3f ms   packed pair test: %.3f ms   ratio %.2f\n", s, p, s / p);
  return 0;
}
