// Issue rate of the packed f32 VALU forms (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two f32 operations per lane and instruction)
// against v_fma_f32 / v_mul_f32 on gfx950, 8 waves per SIMD, 8 independent chains per lane.  The question it answers: does a
// packed instruction cost ONE issue slot (then the slab tests and the Moeller-Trumbore pairs could halve) or two?
// hipcc --offload-arch=gfx950 -O3 -o build/pk_rate tools/micro/pk_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, float seed, int iters) {
  f2 a[8]; float s[8];
  for (int i = 0; i < 8; ++i) { a[i] = (f2){seed + threadIdx.x + i, seed - i}; s[i] = seed + threadIdx.x * 0.5f + i; }
  const f2 c = {1.0000001f, 0.9999999f}, d = {0.5f, 0.25f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));
      else if (MODE == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(c.x), "v"(d.x));
      else if (MODE == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      else if (MODE == 3) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(s[i]) : "v"(c.x));
      else if (MODE == 4) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(d));
      else asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(s[i]));
    }
  }
  float r = 0;
  for (int i = 0; i < 8; ++i) r += a[i].x + a[i].y + s[i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int MODE> double run(float* d, int iters) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(256 * 8), dim3(256), 0, 0, d, 1.0f, 10);
  hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(256 * 8), dim3(256), 0, 0, d, 1.0f, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms;
}
int main() {
  float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
  const int iters = 20000;
  const char* names[6] = {"v_pk_fma_f32", "v_fma_f32", "v_pk_mul_f32", "v_mul_f32", "v_pk_add_f32", "v_cvt_f32_ubyte0"};
  double ms[6] = {run<0>(d, iters), run<1>(d, iters), run<2>(d, iters), run<3>(d, iters), run<4>(d, iters), run<5>(d, iters)};
  for (int m = 0; m < 6; ++m) {
    double instr = (double)256 * 8 * 4 * 8.0 * iters;                      // wave-instructions executed
    printf("%-18s %8.3f ms  %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", names[m], ms[m], 1024 * 2.4e9 * (ms[m] * 1e-3) / instr);
  }
  return 0;
}
