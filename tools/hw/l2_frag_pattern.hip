// Probe: the weight-fragment access pattern of the fused kernels (every CU streams the SAME blocked matrix [K/32][NR][32] x 16 bit;
// wave w of a workgroup reads rows n0(w)..n0(w)+31 of k-block kb as two 1-KiB loads), with different ways of decorrelating the CUs:
//   same   every workgroup: wave w -> rows 32w, k-blocks 0,1,2,...
//   nrot   wave w of workgroup g -> rows 32((w + g') % 8)              (different row blocks at the same time, same k order)
//   krot   k-blocks rotated by g'                                       (different k-blocks at the same time)
//   both
// g' = blockIdx.x >> 3 (blocks b and b + 8 share an XCD).   hipcc --offload-arch=gfx950 -O3 l2_frag_pattern.hip && ./a.out
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void stream(const u32x4* __restrict__ w, int KB, int NR, int chunks, unsigned* sink, int reps) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, fi = lane & 15, fg = lane >> 4;
  const int g = blockIdx.x >> 3;
  const int wrow = (MODE & 1) ? ((wave + g) & 7) : wave;
  const int krot = (MODE & 2) ? (g & (KB - 1)) : 0;
  u32x4 acc = {0, 0, 0, 0};
  for (int r = 0; r < reps; ++r)
    for (int c = 0; c < chunks; ++c) {          // chunk c: rows 256c .. 256c+255 of all KB k-blocks (one "weight set": KB x 16 KiB)
      u32x4 v[2][8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int kb = (j + krot) & (KB - 1);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const size_t row = (size_t)kb * NR + c * 256 + wrow * 32 + nt * 16 + fi;
          v[nt][j] = w[row * 4 + fg];           // 64-byte rows = 4 x 16 B
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) acc ^= v[0][j] ^ v[1][j];
    }
  if (acc[0] == 0x12345678u) sink[0] = acc[1];
}

int main() {
  const int KB = 8, NR = 1024, chunks = 4;                     // linear1 of the FFN: [256/32][1024][32] x 2 B = 512 KiB
  const size_t bytes = (size_t)KB * NR * 64;
  u32x4* w; unsigned* sink;
  (void)hipMalloc(&w, bytes); (void)hipMemset(w, 1, bytes); (void)hipMalloc(&sink, 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto run = [&](auto kern, const char* name) {
    const int reps = 8;
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, w, KB, NR, chunks, sink, reps);
    (void)hipEventRecord(e0);
    for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, w, KB, NR, chunks, sink, reps);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / 10 / reps;
    printf("%-6s %7.2f us per 512 KiB pass -> %6.1f GB/s per CU, %5.2f TB/s chip\n", name, us, bytes / us / 1e3, bytes * 256.0 / us / 1e6);
  };
  run(stream<0>, "same");
  run(stream<1>, "nrot");
  run(stream<2>, "krot");
  run(stream<3>, "both");
  return 0;
}
