// Does a wave64 VALU instruction whose EXEC mask leaves one 32-lane HALF empty issue in one pass on CDNA4's SIMD-32
// (MI355X_MICROARCH.md: "a wave issues each VALU instruction over 2 cycles, 32 lanes per cycle")?  If it did, WHERE the live
// lanes of a diverged wave sit would be worth up to 2x on every instruction that runs at <= 32 lanes -- most of k_path_tree on
// config 5 (26.5 of 64 lanes per VALU instruction) -- and packing the walking rays into one half would pay.
//
// Two instruction streams, 8 waves per SIMD, every CU busy, EXEC set once before the loop with s_mov_b64:
//   fma   eight independent v_fma_f32 chains (hard-wired registers, sources in three VGPR banks: vgpr_bank.hip's mode A)
//   tri   the Moeller-Trumbore mix of tri_pk.hip (tri_test_bf of lr_path.h, compiled code under the same EXEC)
// EXEC patterns: all 64 | low 32 | high 32 | even lanes (32 scattered) | low 16 | lanes 0-7 of each 32 (16 scattered) | one lane.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -I lumillyrender_amd/csrc -o build/exec_half tools/micro/exec_half.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include "lr_math.h"
using namespace lr;

#define CHAINS(OP) OP(8) OP(12) OP(16) OP(20) OP(24) OP(28) OP(32) OP(36)
__global__ void __launch_bounds__(256) k_fma(float* out, float seed, int iters, unsigned long long mask) {
  float r;
  asm volatile(
    "s_mov_b64 s[22:23], exec\n"
    "v_mov_b32 v1, 1.0\n v_mov_b32 v2, 0.5\n"
    "v_mov_b32 v8, %1\n v_mov_b32 v12, %1\n v_mov_b32 v16, %1\n v_mov_b32 v20, %1\n v_mov_b32 v24, %1\n v_mov_b32 v28, %1\n v_mov_b32 v32, %1\n v_mov_b32 v36, %1\n"
    "s_mov_b32 s20, %2\n"
    "s_mov_b64 exec, %3\n"
    "1:\n"
#define A(D) "v_fma_f32 v" #D ", v" #D ", v1, v2\n"
    CHAINS(A) CHAINS(A)
    "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n"
    "s_mov_b64 exec, s[22:23]\n"
    "v_add_f32 %0, v8, v12\n v_add_f32 %0, %0, v16\n v_add_f32 %0, %0, v36\n"
    : "=v"(r) : "v"(seed), "s"(iters), "s"(mask)
    : "v1", "v2", "v8", "v12", "v16", "v20", "v24", "v28", "v32", "v36", "s20", "s22", "s23", "scc", "memory");
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

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
// the lanes of `mask` run the loop, the others skip it: a plain divergent branch, as in the render kernels
__global__ void __launch_bounds__(256) k_tri(float* out, float seed, int iters, unsigned long long mask) {
  V3 pa = v3(seed + threadIdx.x, 1, 2), e1a = v3(1, seed, 0.5f), e2a = v3(0.25f, 1, seed);
  V3 o = v3(0.1f * threadIdx.x, 0.2f, 0.3f), d = v3(0.3f, 0.5f, 0.8f);
  float best = 3e38f;
  if ((mask >> (threadIdx.x & 63u)) & 1ull) {
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
      float tA; bool hA = tri_test_bf(pa, e1a, e2a, o, d, &tA);
      best = (hA && tA < best) ? tA : best;                  // a select, not a branch: EXEC stays what the outer `if` made it
      o.x += 1e-3f; asm volatile("" : "+v"(pa.x), "+v"(o.y));
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = best;
}

template <class K> double run(K kern, float* d, int iters, unsigned long long mask) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL(kern, dim3(256 * 8), dim3(256), 0, 0, d, 1.5f, 10, mask);
  (void)hipDeviceSynchronize();
  double best = 1e30;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(256 * 8), dim3(256), 0, 0, d, 1.5f, iters, mask);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  return best;
}
int main() {
  float* d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
  struct { const char* name; unsigned long long m; } pat[] = {
    {"all 64 lanes", ~0ull}, {"low 32 lanes", 0x00000000ffffffffull}, {"high 32 lanes", 0xffffffff00000000ull},
    {"even lanes (32 scattered)", 0x5555555555555555ull}, {"low 16 lanes", 0xffffull}, {"lanes 0-7 of each 32 (16 scattered)", 0x000000ff000000ffull},
    {"lanes 16-47 (32, straddling the halves)", 0x0000ffffffff0000ull}, {"one lane", 1ull}};
  const int it_fma = 10000, it_tri = 4000;
  double f0 = 0, t0 = 0;
  printf("%-42s %12s %8s %12s %8s\n", "EXEC", "fma ms", "vs all", "tri ms", "vs all");
  for (auto& p : pat) {
    double f = run(k_fma, d, it_fma, p.m), t = run(k_tri, d, it_tri, p.m);
    if (!f0) { f0 = f; t0 = t; }
    printf("%-42s %12.3f %8.3f %12.3f %8.3f\n", p.name, f, f / f0, t, t / t0);
  }
  printf("fma stream: %.2f cycles per wave-instruction per SIMD at 2.4 GHz with all lanes\n", 1024 * 2.4e9 * (f0 * 1e-3) / ((double)256 * 8 * 4 * 16.0 * it_fma));
  return 0;
}
