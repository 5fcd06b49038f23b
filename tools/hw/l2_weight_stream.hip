// Probe: how fast can every CU stream the SAME weight buffer (L2-resident) into registers?
// grid = 256 workgroups x 512 threads; each wave reads its 1/8 slice of a `bytes` buffer as 1-KiB wave loads, DEPTH in flight.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH, bool ROT>
__global__ __launch_bounds__(512) void stream(const u32x4* __restrict__ w, size_t n16 /* 16-byte elements */, unsigned* sink, int reps) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t per_wave = n16 / 8;              // elements per wave slice
  const size_t nload = per_wave / 64;           // wave-loads per slice
  const size_t rot = ROT ? ((blockIdx.x >> 3) & 31) * (nload / 32) : 0;
  u32x4 acc = {0, 0, 0, 0};
  for (int r = 0; r < reps; ++r) {
    for (size_t i = 0; i < nload; i += DEPTH) {
      u32x4 v[DEPTH];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        size_t j = (i + d + rot) % nload;
        v[d] = w[wave * per_wave + j * 64 + lane];
      }
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) acc ^= v[d];
    }
  }
  if (acc[0] == 0x12345678u) sink[0] = acc[1];
}

int main() {
  const size_t bytes = 1 << 20;
  u32x4* w; unsigned* sink;
  (void)hipMalloc(&w, bytes); (void)hipMemset(w, 1, bytes); (void)hipMalloc(&sink, 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto run = [&](auto kern, const char* name, int grid) {
    const int reps = 4;
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 0, 0, w, bytes / 16, sink, reps);
    (void)hipEventRecord(e0);
    for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 0, 0, w, bytes / 16, sink, reps);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double us = ms * 1e3 / 10 / reps;
    printf("%-28s grid %4d: %7.2f us per 1 MiB pass  -> %6.1f GB/s per CU, %5.2f TB/s chip\n", name, grid, us, bytes / us / 1e3,
           bytes * (double)grid / us / 1e6);
  };
  run(stream<8, false>, "depth 8", 256);
  run(stream<16, false>, "depth 16", 256);
  run(stream<32, false>, "depth 32", 256);
  run(stream<32, true>, "depth 32 rotated", 256);
  run(stream<16, true>, "depth 16 rotated", 256);
  run(stream<32, true>, "depth 32 rotated 2 WG/CU", 512);
  // does the per-CU rate depend on how many CUs stream at the same time?  (a shared L2 limit would, a per-CU limit would not)
  for (int g : {8, 32, 64, 128, 192, 256}) run(stream<16, false>, "depth 16, fewer workgroups", g);
  return 0;
}
