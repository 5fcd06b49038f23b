// Probe: do the three streams of an FFN GEMM phase overlap on one CU?  A phase of the N-split kernels is, per wave and k-step:
// 2 weight-fragment loads (1 KiB each, L2 -> VGPR, for the NEXT phase), 4 activation-fragment reads from LDS (ds_read_b128) and
// 8 MFMAs (16x16x32 f16); 8 k-steps per phase, 8 waves per workgroup, one workgroup per CU.  Each stream can be switched off
// (its operand registers then hold constants) or, for the weights, pointed at one L1-resident KiB.
//   W: 0 none | 1 fragment pattern from L2 (1 MiB, what the kernels do) | 2 linear 1-KiB wave loads from L2 | 3 one L1-resident KiB
//   D: LDS reads on / off        M: MFMAs on / off        R: rows per phase = 64 * R (R = 2: every weight fragment multiplies two halves)
// hipcc --offload-arch=gfx950 -O3 -o ffn_overlap ffn_overlap.hip && ./ffn_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f16x8 as_h(u32x4 v) { return __builtin_bit_cast(f16x8, v); }

template <int W, int D, int M, int R>
__device__ __forceinline__ void one_phase(const u32x4* __restrict__ w, const char* lds, u32x4 (&cur)[2][8], u32x4 (&nxt)[2][8], f32x4 (&acc)[2][4],
                                          int set, int wave, int lane, int fi, int fg) {
  const u32x4 bconst = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
#pragma unroll
  for (int h = 0; h < R; ++h) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      u32x4 b[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        if (D) {
          const int row = mt * 16 + fi;
          const int c = fg ^ ((4 - ((row >> 2) & 3)) & 3);
          b[mt] = *reinterpret_cast<const u32x4*>(lds + h * 32768 + ((j * 64 + row) * 4 + c) * 16);
        } else b[mt] = bconst;
      }
      if (W && h == R - 1) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          size_t idx;
          if (W == 1) idx = ((size_t)(set * 8 + j) * 256 + wave * 32 + nt * 16 + fi) * 4 + fg;      // [kb][256 rows][64 B]
          else if (W == 2) idx = ((size_t)(set * 8 + j) * 256 + wave * 32 + nt * 16) * 4 + lane;     // the same KiB, lanes linear
          else idx = (size_t)(wave * 2 + nt) * 64 + lane;                                            // always the same 16 KiB per CU
          nxt[nt][j] = w[idx];
        }
      }
      if (M) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
            acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h(cur[nt][j]), as_h(b[mt]), acc[nt][mt], 0, 0, 0);
      } else {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) acc[nt][mt][0] += __builtin_bit_cast(float, cur[nt][j][0] ^ b[mt][nt]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __syncthreads();       // the phase boundary of the kernels (activation epilogue + barrier): here just the barrier
}

template <int W, int D, int M, int R>
__global__ __launch_bounds__(512) void phase_kernel(const u32x4* __restrict__ w, float* sink, int phases) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, fi = lane & 15, fg = lane >> 4;
  for (int i = tid; i < R * 32768 / 16; i += 512) reinterpret_cast<u32x4*>(lds)[i] = u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
  __syncthreads();
  u32x4 wa[2][8], wb[2][8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) wa[nt][j] = wb[nt][j] = u32x4{0x2c002c00u, 0x2c002c00u, 0x2c002c00u, 0x2c002c00u};
  f32x4 acc[2][4] = {};
  for (int p = 0; p < phases; p += 2) {
    one_phase<W, D, M, R>(w, lds, wa, wb, acc, p & 7, wave, lane, fi, fg);
    one_phase<W, D, M, R>(w, lds, wb, wa, acc, (p + 1) & 7, wave, lane, fi, fg);
  }
  float s = 0.f;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) s += acc[nt][mt][0] + acc[nt][mt][1] + acc[nt][mt][2] + acc[nt][mt][3];
  if (s == 123.456f) sink[0] = s;
}

int main() {
  const size_t bytes = 1 << 20;
  u32x4* w; float* sink;
  (void)hipMalloc(&w, bytes); (void)hipMemset(w, 0x2c, bytes); (void)hipMalloc(&sink, 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  int clk = 0; (void)hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
  printf("per PHASE of one workgroup per CU (256 workgroups), clock attribute %d kHz\n", clk);
  auto run = [&](auto kern, const char* name, int R) {
    const int phases = 64;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, R * 32768);
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(kern, dim3(256), dim3(512), R * 32768, 0, w, sink, phases);
    (void)hipEventRecord(e0);
    for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(kern, dim3(256), dim3(512), R * 32768, 0, w, sink, phases);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / 10 / phases;
    printf("%-58s %6.3f us per phase  (%5.0f cycles at 2.1 GHz)\n", name, us, us * 2100);
  };
  run(phase_kernel<0, 0, 1, 1>, "MFMA only", 1);
  run(phase_kernel<0, 1, 0, 1>, "LDS reads only", 1);
  run(phase_kernel<0, 1, 1, 1>, "MFMA + LDS", 1);
  run(phase_kernel<1, 0, 0, 1>, "weights (fragment pattern, L2) only", 1);
  run(phase_kernel<2, 0, 0, 1>, "weights (linear lanes, L2) only", 1);
  run(phase_kernel<3, 0, 0, 1>, "weights (one L1-resident KiB per load) only", 1);
  run(phase_kernel<1, 0, 1, 1>, "weights (fragment, L2) + MFMA", 1);
  run(phase_kernel<1, 1, 0, 1>, "weights (fragment, L2) + LDS", 1);
  run(phase_kernel<1, 1, 1, 1>, "weights (fragment, L2) + LDS + MFMA   [the kernels]", 1);
  run(phase_kernel<2, 1, 1, 1>, "weights (linear, L2) + LDS + MFMA", 1);
  run(phase_kernel<3, 1, 1, 1>, "weights (L1) + LDS + MFMA", 1);
  run(phase_kernel<0, 1, 1, 2>, "128 rows: MFMA + LDS", 2);
  run(phase_kernel<1, 1, 1, 2>, "128 rows: weights (fragment, L2) + LDS + MFMA", 2);
  run(phase_kernel<2, 1, 1, 2>, "128 rows: weights (linear, L2) + LDS + MFMA", 2);
  return 0;
}
