// Do the packed f32 VALU forms (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) share the issue port of the plain VALU, or run beside it?
// Three kernels of independent chains, 8 waves per SIMD:  X = N plain v_fma_f32;  Y = N v_pk_fma_f32;  Z = N plain + N packed interleaved.
// Same port: t(Z) ~ t(X) + t(Y).  Separate ports: t(Z) ~ max.  Also: packed forms with an SGPR pair as a broadcast operand (op_sel),
// and a bit-equality check of v_pk_mul_f32 / v_pk_add_f32 against v_mul_f32 / v_add_f32 on random operands (parity depends on it).
// hipcc --offload-arch=gfx950 -O3 -o build/pk_port tools/micro/pk_port.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, float seed, int iters, f2 sc) {
  f2 a[6]; float s[6];
  for (int i = 0; i < 6; ++i) { a[i] = (f2){seed + threadIdx.x + i, seed - i}; s[i] = seed + threadIdx.x * 0.5f + i; }
  const f2 c = {1.0000001f, 0.9999999f}, d = {0.5f, 0.25f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      if (MODE == 0 || MODE == 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(c.x), "v"(d.x));
      if (MODE == 1 || MODE == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));
      if (MODE == 3) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(a[i]) : "s"(sc));          // both halves times sc.x (SGPR broadcast)
      if (MODE == 4) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(s[i]) : "s"(sc.x));
      if (MODE == 5) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(s[i]) : "v"(c.x)); }
      if (MODE == 6) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(s[i]) : "v"(c.x)); asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d)); }
    }
  }
  float r = 0;
  for (int i = 0; i < 6; ++i) r += a[i].x + a[i].y + s[i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
__global__ void k_bits(const float* x, const float* y, uint32_t* bad, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 >= n) return;
  f2 a = {x[2 * i], x[2 * i + 1]}, b = {y[2 * i], y[2 * i + 1]}, pm, pa, pf;
  asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(pm) : "v"(a), "v"(b));
  asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(pa) : "v"(a), "v"(b));
  asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(pf) : "v"(a), "v"(b), "v"(pa));
  float m0, m1, a0, a1, f0, f1;
  asm volatile("v_mul_f32 %0, %1, %2" : "=v"(m0) : "v"(a.x), "v"(b.x)); asm volatile("v_mul_f32 %0, %1, %2" : "=v"(m1) : "v"(a.y), "v"(b.y));
  asm volatile("v_add_f32 %0, %1, %2" : "=v"(a0) : "v"(a.x), "v"(b.x)); asm volatile("v_add_f32 %0, %1, %2" : "=v"(a1) : "v"(a.y), "v"(b.y));
  asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(f0) : "v"(a.x), "v"(b.x), "v"(a0)); asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(f1) : "v"(a.y), "v"(b.y), "v"(a1));
  auto ne = [](float p, float q) { return __float_as_uint(p) != __float_as_uint(q) && !(p != p && q != q); };
  if (ne(pm.x, m0) || ne(pm.y, m1)) atomicAdd(&bad[0], 1u);
  if (ne(pa.x, a0) || ne(pa.y, a1)) atomicAdd(&bad[1], 1u);
  if (ne(pf.x, f0) || ne(pf.y, f1)) atomicAdd(&bad[2], 1u);
}
template <int MODE> double run(float* d, int iters) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  const f2 sc = {1.0000001f, 0.5f};
  hipLaunchKernelGGL(k<MODE>, dim3(256 * 8), dim3(256), 0, 0, d, 1.0f, 10, sc);
  (void)hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(256 * 8), dim3(256), 0, 0, d, 1.0f, iters, sc);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  return ms;
}
int main() {
  float* d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
  const int iters = 20000;
  const char* names[7] = {"X: v_fma_f32", "Y: v_pk_fma_f32", "Z: X + Y interleaved", "v_pk_mul_f32 sgpr-bcast", "v_mul_f32 sgpr", "C: v_cndmask_b32", "C + Y interleaved"};
  double ms[7] = {run<0>(d, iters), run<1>(d, iters), run<2>(d, iters), run<3>(d, iters), run<4>(d, iters), run<5>(d, iters), run<6>(d, iters)};
  for (int m = 0; m < 7; ++m) printf("%-26s %8.3f ms\n", names[m], ms[m]);
  printf("same port predicts Z = X + Y = %.3f ms, separate ports max = %.3f ms; measured %.3f ms\n", ms[0] + ms[1], ms[0] > ms[1] ? ms[0] : ms[1], ms[2]);
  printf("C + Y: sum %.3f ms, max %.3f ms, measured %.3f ms\n", ms[5] + ms[1], ms[5] > ms[1] ? ms[5] : ms[1], ms[6]);
  // bit equality on 2^24 random operand pairs over the whole exponent range (denormals included)
  const int n = 1 << 24;
  std::vector<float> hx(n), hy(n);
  uint64_t st = 88172645463325252ull;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (uint32_t)(st >> 16); };
  for (int i = 0; i < n; ++i) { uint32_t u = rnd(), v = rnd(); if ((i & 7) == 0) { u &= 0x807fffffu; } memcpy(&hx[i], &u, 4); memcpy(&hy[i], &v, 4); }
  float *dx, *dy; uint32_t* bad; (void)hipMalloc(&dx, n * 4); (void)hipMalloc(&dy, n * 4); (void)hipMalloc(&bad, 16);
  (void)hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dy, hy.data(), n * 4, hipMemcpyHostToDevice); (void)hipMemset(bad, 0, 16);
  hipLaunchKernelGGL(k_bits, dim3(n / 2 / 256), dim3(256), 0, 0, dx, dy, bad, n);
  uint32_t hb[4]; (void)hipMemcpy(hb, bad, 16, hipMemcpyDeviceToHost);
  printf("bit mismatches packed vs plain on %d pairs: mul %u  add %u  fma %u\n", n / 2, hb[0], hb[1], hb[2]);
  return 0;
}
