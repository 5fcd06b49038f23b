// Probe: what a dependent chain of small launches costs per launch, against the dynamic LDS size and the code size of the kernel.
//   hipcc --offload-arch=gfx950 -O3 -o launch_cost launch_cost.hip && ./launch_cost
#include <hip/hip_runtime.h>
#include <cstdio>

extern __shared__ float smem[];
__global__ __launch_bounds__(512) void k_small(float* p, int n) {
  if (n > 0) smem[threadIdx.x] = p[threadIdx.x];
  __syncthreads();
  if (threadIdx.x == 0 && n > 0) p[blockIdx.x] = smem[1] + 1.f;
}
// the same with a long (never executed at run time, but resident in the code object) unrolled body: code size ~ REP * 8 instructions
template <int REP>
__global__ __launch_bounds__(512) void k_big(float* p, int n) {
  float a = p[threadIdx.x];
  if (n > 1000000) {
#pragma unroll
    for (int i = 0; i < REP; ++i) a = a * 1.0001f + p[(threadIdx.x + i) & 1023];
  }
  if (threadIdx.x == 0) p[blockIdx.x] = a;
}

template <class F>
double chain(F launch, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3 / reps;
}

int main() {
  float* p; hipMalloc(&p, 1 << 20); hipMemset(p, 0, 1 << 20);
  hipFuncSetAttribute((const void*)k_small, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int wgs : {8, 64, 256, 1024})
    for (int lds : {0, 16 * 1024, 64 * 1024, 96 * 1024, 152 * 1024}) {
      double us = chain([&] { hipLaunchKernelGGL(k_small, dim3(wgs), dim3(512), lds, 0, p, 1); }, 500);
      printf("k_small  %5d workgroups  %3d KiB dynamic LDS: %6.2f us per dependent launch\n", wgs, lds / 1024, us);
    }
  printf("k_big<64>   8 workgroups: %6.2f us\n", chain([&] { hipLaunchKernelGGL(k_big<64>, dim3(8), dim3(512), 0, 0, p, 1); }, 500));
  printf("k_big<1024> 8 workgroups: %6.2f us\n", chain([&] { hipLaunchKernelGGL(k_big<1024>, dim3(8), dim3(512), 0, 0, p, 1); }, 500));
  printf("k_big<4096> 8 workgroups: %6.2f us\n", chain([&] { hipLaunchKernelGGL(k_big<4096>, dim3(8), dim3(512), 0, 0, p, 1); }, 500));
  printf("%s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
