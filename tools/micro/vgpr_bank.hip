// Do two VGPR source operands in the same register bank (index mod 4) cost a wave64 VALU instruction extra issue time on gfx950?
// Eight independent v_fma_f32 chains per wave with hard-wired registers, 8 waves per SIMD:
//   A  sources in three different banks      v_fma_f32 vD, vD, v1, v2      (D = 8, 12, ..: bank 0; v1: bank 1; v2: bank 2)
//   B  two sources in one bank               v_fma_f32 vD, vD, v1, v5      (v1, v5: bank 1)
//   C  three sources in one bank             v_fma_f32 vD, vD, v4, v40     (vD, v4, v40: bank 0)
//   D  VOP2 form, two sources in one bank    v_mul_f32 vD, vD, v4
//   E  VOP2 form, different banks            v_mul_f32 vD, vD, v1
// hipcc --offload-arch=gfx950 -O3 -o build/vgpr_bank tools/micro/vgpr_bank.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHAINS(OP) \
  OP(8) OP(12) OP(16) OP(20) OP(24) OP(28) OP(32) OP(36)
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, float seed, int iters) {
  float r;
  asm volatile(
    "v_mov_b32 v1, 1.0\n v_mov_b32 v2, 0.5\n v_mov_b32 v5, 0.5\n v_mov_b32 v4, 1.0\n v_mov_b32 v40, 0.5\n"
    "v_mov_b32 v8, %1\n v_mov_b32 v12, %1\n v_mov_b32 v16, %1\n v_mov_b32 v20, %1\n v_mov_b32 v24, %1\n v_mov_b32 v28, %1\n v_mov_b32 v32, %1\n v_mov_b32 v36, %1\n"
    "s_mov_b32 s20, %2\n"
    "1:\n"
#define A(D) "v_fma_f32 v" #D ", v" #D ", v1, v2\n"
#define B(D) "v_fma_f32 v" #D ", v" #D ", v1, v5\n"
#define C(D) "v_fma_f32 v" #D ", v" #D ", v4, v40\n"
#define DD(D) "v_mul_f32 v" #D ", v" #D ", v4\n"
#define E(D) "v_mul_f32 v" #D ", v" #D ", v1\n"
    ".if %3 == 0\n" CHAINS(A) ".endif\n"
    ".if %3 == 1\n" CHAINS(B) ".endif\n"
    ".if %3 == 2\n" CHAINS(C) ".endif\n"
    ".if %3 == 3\n" CHAINS(DD) ".endif\n"
    ".if %3 == 4\n" CHAINS(E) ".endif\n"
    "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n"
    "v_add_f32 %0, v8, v12\n v_add_f32 %0, %0, v16\n v_add_f32 %0, %0, v36\n"
    : "=v"(r) : "v"(seed), "s"(iters), "n"(MODE)
    : "v1", "v2", "v4", "v5", "v8", "v12", "v16", "v20", "v24", "v28", "v32", "v36", "v40", "s20", "scc", "memory");
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int MODE> double run(float* d, int iters) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(256 * 8), dim3(256), 0, 0, d, 1.0f, 10);
  (void)hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(256 * 8), dim3(256), 0, 0, d, 1.0f, iters);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  return ms;
}
int main() {
  float* d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
  const int iters = 20000;
  const char* names[5] = {"fma, three banks", "fma, two sources in one bank", "fma, three sources in one bank", "mul (VOP2), both in one bank", "mul (VOP2), two banks"};
  double ms[5] = {run<0>(d, iters), run<1>(d, iters), run<2>(d, iters), run<3>(d, iters), run<4>(d, iters)};
  for (int m = 0; m < 5; ++m) printf("%-34s %8.3f ms  %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", names[m], ms[m], 1024 * 2.4e9 * (ms[m] * 1e-3) / ((double)256 * 8 * 4 * 8.0 * iters));
  return 0;
}
