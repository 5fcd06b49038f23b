// Probe: is "MFMA result -> VALU read" (compiler-inserted s_nop only) safe when the SIMD's partner wave is MFMA-dense?
// waves 0-3 (one per SIMD): repeat { chain of NCH MFMAs ; VALU reads the result right away ; compare with expected }
// waves 4-7: dense MFMA loop (partner load), or idle when partner==0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NCH>
__global__ __launch_bounds__(512) void probe(int iters, int partner, unsigned* bad, float* sink) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)1.0f; b[i] = (_Float16)1.0f; }
  if (wave >= 4) {
    if (!partner) return;
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int it = 0; it < iters * 8; ++it) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[j], 0, 0, 0);
    }
    sink[blockIdx.x * 512 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    return;
  }
  unsigned nbad = 0;
  float tot = 0.f;
  for (int it = 0; it < iters; ++it) {
    f32x4 acc = {(float)it, (float)it, (float)it, (float)it};
#pragma unroll
    for (int j = 0; j < NCH; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    // expected: it + 32*NCH in every element
    float v = acc[0] + 0.5f;                       // VALU read straight after the MFMA (compiler pads s_nop)
    if (v != (float)it + 32.f * NCH + 0.5f) ++nbad;
    if (acc[3] != (float)it + 32.f * NCH) ++nbad;
    tot += v;
    asm volatile("" : "+v"(a), "+v"(b));
  }
  if (nbad) atomicAdd(bad, nbad);
  sink[blockIdx.x * 512 + threadIdx.x] = tot;
}

int main() {
  unsigned* bad; float* sink;
  hipMalloc(&bad, 4); hipMalloc(&sink, 1024 * 512 * 4);
  for (int partner = 0; partner < 2; ++partner) {
    hipMemset(bad, 0, 4);
    hipLaunchKernelGGL((probe<1>), dim3(1024), dim3(512), 0, 0, 20000, partner, bad, sink);
    hipLaunchKernelGGL((probe<2>), dim3(1024), dim3(512), 0, 0, 20000, partner, bad, sink);
    hipLaunchKernelGGL((probe<8>), dim3(1024), dim3(512), 0, 0, 20000, partner, bad, sink);
    hipDeviceSynchronize();
    unsigned h; hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
    printf("partner=%d mismatches=%u  (%s)\n", partner, h, hipGetErrorString(hipGetLastError()));
  }
  return 0;
}
