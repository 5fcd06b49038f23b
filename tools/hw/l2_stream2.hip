// Probe (round 5): what limits the per-CU L2 -> CU weight stream (~ 35 - 40 B/clk in the kernels, 64 B/clk on paper)?
// Every CU streams the SAME 1 MiB buffer (L2-resident), 8 or 16 waves per CU, a rolling window of DEPTH 1-KiB wave loads per wave,
// with different instruction forms:  plain global_load_dwordx4 | nontemporal | buffer_load (plain / sc1 / nt) | LDS-DMA (global_load_lds_dwordx4).
// hipcc --offload-arch=gfx950 -O3 -o l2_stream2 l2_stream2.hip && ./l2_stream2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

enum { PLAIN = 0, NT = 1, BUF = 2, BUF_SC1 = 3, BUF_NT = 4, LDSDMA = 5 };

template <int MODE, int DEPTH, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void stream(const u32x4* __restrict__ w, size_t n16, unsigned* sink, int reps) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t per_wave = n16 / WAVES;          // 16-byte elements per wave slice
  const int nload = (int)(per_wave / 64);       // wave-loads per slice
  const u32x4* base = w + wave * per_wave + lane;
  u32x4 acc = {0, 0, 0, 0};
  if constexpr (MODE == LDSDMA) {
    // ring of DEPTH 1-KiB slots per wave in LDS; wait for the oldest with a counted vmcnt, never read (rate test)
    char* myl = lds + wave * DEPTH * 1024;
    for (int r = 0; r < reps; ++r) {
      for (int i = 0; i < nload; i += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
          __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(base + (size_t)(i + d) * 64),
                                           (void __attribute__((address_space(3)))*)(myl + d * 1024), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH / 2) : "memory");
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc[0] = *reinterpret_cast<unsigned*>(myl + lane * 4);
  } else {
    [[maybe_unused]] __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(w), 0, -1, 0x00020000);
    auto ld = [&](int j) -> u32x4 {
      if constexpr (MODE == PLAIN) return base[(size_t)j * 64];
      else if constexpr (MODE == NT) return __builtin_nontemporal_load(base + (size_t)j * 64);
      else {
        const unsigned off = (unsigned)((wave * per_wave + (size_t)j * 64 + lane) * 16);
        constexpr int aux = MODE == BUF ? 0 : MODE == BUF_SC1 ? 16 : 2;
        return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, aux));
      }
    };
    u32x4 v[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) v[d] = ld(d);
    for (int r = 0; r < reps; ++r) {
      for (int i = 0; i < nload; i += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
          acc ^= v[d];                               // consume the oldest, re-request its slot DEPTH loads ahead
          int j = i + DEPTH + d;
          if (j >= nload) j -= nload;
          v[d] = ld(j);
        }
      }
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) acc ^= v[d];
  }
  if (acc[0] == 0x12345678u) sink[0] = acc[1];
}

int main() {
  const size_t bytes = 1 << 20;
  u32x4* w; unsigned* sink;
  (void)hipMalloc(&w, bytes); (void)hipMemset(w, 1, bytes); (void)hipMalloc(&sink, 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto run = [&](auto kern, const char* name, int grid, int threads, size_t lds) {
    const int reps = 8;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, 0, w, bytes / 16, sink, reps);
    (void)hipEventRecord(e0);
    for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, 0, w, bytes / 16, sink, reps);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / 10 / reps;
    printf("%-52s grid %4d x %4d: %7.2f us per 1 MiB pass -> %6.1f GB/s per workgroup (%5.1f B/clk at 2.1 GHz), %5.2f TB/s chip\n", name, grid, threads, us,
           bytes / us / 1e3, bytes / us / 1e3 / 2.1, bytes * (double)grid / us / 1e6);
  };
  run(stream<PLAIN, 8, 8>, "global_load_dwordx4, window 8", 256, 512, 0);
  run(stream<PLAIN, 16, 8>, "global_load_dwordx4, window 16", 256, 512, 0);
  run(stream<PLAIN, 32, 8>, "global_load_dwordx4, window 32", 256, 512, 0);
  run(stream<NT, 16, 8>, "global_load_dwordx4 nt, window 16", 256, 512, 0);
  run(stream<BUF, 16, 8>, "buffer_load_dwordx4, window 16", 256, 512, 0);
  run(stream<BUF_SC1, 16, 8>, "buffer_load_dwordx4 sc1, window 16", 256, 512, 0);
  run(stream<BUF_NT, 16, 8>, "buffer_load_dwordx4 nt, window 16", 256, 512, 0);
  run(stream<LDSDMA, 8, 8>, "global_load_lds_dwordx4 (LDS-DMA), window 8", 256, 512, 8 * 8 * 1024);
  run(stream<LDSDMA, 16, 8>, "global_load_lds_dwordx4 (LDS-DMA), window 16", 256, 512, 8 * 16 * 1024);
  run(stream<PLAIN, 16, 16>, "global_load_dwordx4, window 16, 16 waves", 256, 1024, 0);
  run(stream<PLAIN, 8, 16>, "global_load_dwordx4, window 8, 16 waves", 256, 1024, 0);
  run(stream<PLAIN, 16, 4>, "global_load_dwordx4, window 16, 4 waves", 256, 256, 0);
  run(stream<PLAIN, 16, 8>, "global_load_dwordx4, window 16, 1 workgroup per XCD", 8, 512, 0);
  run(stream<LDSDMA, 8, 8>, "LDS-DMA, window 8, 1 workgroup per XCD", 8, 512, 8 * 8 * 1024);
  return 0;
}
