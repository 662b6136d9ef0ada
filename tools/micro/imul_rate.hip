// Issue rate of v_mul_lo_u32 (pcg4d's multiplies) against v_mul_f32 / v_mad_u32_u24 on gfx950: N dependent-free chains per lane,
// 8 waves per SIMD, enough to saturate the VALU port.  hipcc --offload-arch=gfx950 -O3 -o build/imul_rate tools/micro/imul_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t seed, int iters) {
  uint32_t a[8]; float f[8];
  for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 17u + i; f[i] = (float)a[i] * 1e-3f; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) a[i] = a[i] * 1664525u + 1013904223u;                 // v_mul_lo_u32 (+ add) or v_mad_u64_u32
      else if (MODE == 1) f[i] = f[i] * 1.0000001f + 0.5f;                  // v_mul + v_add (no contraction) or fma
      else if (MODE == 2) a[i] = __umul24(a[i], 1664525u) + 1013904223u;    // v_mad_u32_u24
      else a[i] = a[i] * a[(i + 1) & 7];                                    // v_mul_lo_u32, register operands
    }
  }
  uint32_t r = 0;
  for (int i = 0; i < 8; ++i) r ^= a[i] ^ __float_as_uint(f[i]);
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int MODE> double run(uint32_t* d, int iters) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(256 * 8), dim3(256), 0, 0, d, 1u, 10);
  hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(256 * 8), dim3(256), 0, 0, d, 1u, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms;
}
int main() {
  uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
  const int iters = 20000;
  const char* names[4] = {"x*c+k (u32)", "f*c+k (f32, 2 ops)", "mad_u32_u24", "x*y (u32)"};
  double ms[4] = {run<0>(d, iters), run<1>(d, iters), run<2>(d, iters), run<3>(d, iters)};
  for (int m = 0; m < 4; ++m) {
    double ops = (double)256 * 8 * 256 * 8.0 * iters;                      // statements executed, all lanes
    printf("%-22s %8.3f ms  %7.1f G statements/s  (%.2f cycles per wave-statement per SIMD at 2.4 GHz)\n", names[m], ms[m], ops / ms[m] / 1e6, 1024 * 2.4e9 / (ops / 64 / (ms[m] * 1e-3)));
  }
  return 0;
}
