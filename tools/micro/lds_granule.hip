// How many 256-thread workgroups with a given LDS size are resident per CU on gfx950?  Every workgroup spins a fixed time; with
// B workgroups per CU launched, the kernel takes one spin when all B are resident and two when one of them has to wait.  Prints, per
// LDS size, the measured time in spins for B = 5, 6, 7 beside what hipOccupancyMaxActiveBlocksPerMultiprocessor answers.
// hipcc --offload-arch=gfx950 -O3 -o build/lds_granule tools/micro/lds_granule.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256, 7) spin(unsigned long long cycles, unsigned* out) {
  extern __shared__ unsigned lds[];
  lds[threadIdx.x] = threadIdx.x;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) out[blockIdx.x] = lds[0];
}
int main() {
  hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  unsigned* d; hipMalloc(&d, cus * 8 * sizeof(unsigned));
  hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  const unsigned long long cycles = 20000000ull;                      // s_memtime ticks at 100 MHz: 200 ms
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  auto run = [&](int per_cu, size_t lds) {
    hipEventRecord(a);
    hipLaunchKernelGGL(spin, dim3(cus * per_cu), dim3(256), lds, 0, cycles / 100, d);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
  };
  const float unit = run(1, 1024);
  printf("%d CUs, one spin = %.3f ms\n", cus, unit);
  const size_t sizes[] = {20480, 21760, 22624, 23040, 23041, 23405, 25056, 25600, 26080, 26880, 26881, 27104, 27136, 27306, 32768};
  for (size_t lds : sizes) {
    int api = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, (const void*)spin, 256, lds);
    printf("LDS %6zu B: runtime says %d per CU; spins for 5 / 6 / 7 / 8 per CU launched: %.2f %.2f %.2f %.2f\n", lds, api, run(5, lds) / unit, run(6, lds) / unit, run(7, lds) / unit, run(8, lds) / unit);
  }
  return 0;
}
