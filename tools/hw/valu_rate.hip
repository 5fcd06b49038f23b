// Probe: issue cost (shader cycles per wave-instruction) of the VALU instructions the softmax / temporal phases are made of, with
// one wave per SIMD (256 threads) and two (512 threads) -- the fused trajectory kernels run two.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
// Each body is N independent instructions on distinct registers inside one asm statement, repeated in a loop; s_memtime around it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP8(x) x x x x x x x x
#define BODY(NAME, INSTR)                                                                              \
  __global__ __launch_bounds__(512) void NAME(int iters, unsigned long long* out, float* sink) {       \
    float a0 = threadIdx.x * 0.001f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;    \
    float b0 = 1.5f, b1 = 0.5f;                                                                        \
    unsigned long long t0, t1;                                                                         \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");                        \
    for (int i = 0; i < iters; ++i) {                                                                  \
      asm volatile(REP8(INSTR) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));  \
    }                                                                                                  \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");                        \
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;                   \
    sink[blockIdx.x * 512 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                      \
  }

// 8 instructions per REP8 unit x 8 units = 64 instructions per loop iteration... (each unit below is 8 instructions on a0..a7)
#define U8(op, tail) op " %0, %0" tail "\n\t" op " %1, %1" tail "\n\t" op " %2, %2" tail "\n\t" op " %3, %3" tail "\n\t" \
                     op " %4, %4" tail "\n\t" op " %5, %5" tail "\n\t" op " %6, %6" tail "\n\t" op " %7, %7" tail "\n\t"

BODY(k_add, U8("v_add_f32", ", %8"))
BODY(k_sub, U8("v_sub_f32", ", %8"))
BODY(k_exp, U8("v_exp_f32", ""))
BODY(k_rcp, U8("v_rcp_f32", ""))
BODY(k_max, U8("v_max_f32", ", %8"))
BODY(k_max3, U8("v_max3_f32", ", %8, %9"))
BODY(k_maximum3, U8("v_maximum3_f32", ", %8, %9"))
BODY(k_fma, U8("v_fma_f32", ", %8, %9"))
BODY(k_cvtpk, U8("v_cvt_pk_f16_f32", ", %8"))
BODY(k_pkfmah, U8("v_pk_fma_f16", ", %8, %9"))
BODY(k_pkmaxh, U8("v_pk_max_f16", ", %8"))
BODY(k_dot2, U8("v_dot2_f32_f16", ", %8, %9"))
BODY(k_dot2c, U8("v_dot2c_f32_f16", ", %8"))
BODY(k_mov, U8("v_mov_b32", ""))
BODY(k_exph, U8("v_exp_f16", ""))

// packed f32 ops need register pairs
__global__ __launch_bounds__(512) void k_pkadd(int iters, unsigned long long* out, float* sink) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 a0 = {threadIdx.x * 0.001f, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, b = {0.5f, 0.25f};
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int i = 0; i < iters; ++i) {
    asm volatile(REP8("v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %4\n\tv_pk_add_f32 %2, %2, %4\n\tv_pk_add_f32 %3, %3, %4\n\t"
                      "v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %4\n\tv_pk_add_f32 %2, %2, %4\n\tv_pk_add_f32 %3, %3, %4\n\t")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
  sink[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[1] + a2[0] + a3[1];
}
__global__ __launch_bounds__(512) void k_pkmul(int iters, unsigned long long* out, float* sink) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 a0 = {threadIdx.x * 0.001f, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, b = {1.0001f, 0.9999f};
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int i = 0; i < iters; ++i) {
    asm volatile(REP8("v_pk_mul_f32 %0, %0, %4\n\tv_pk_mul_f32 %1, %1, %4\n\tv_pk_mul_f32 %2, %2, %4\n\tv_pk_mul_f32 %3, %3, %4\n\t"
                      "v_pk_mul_f32 %0, %0, %4\n\tv_pk_mul_f32 %1, %1, %4\n\tv_pk_mul_f32 %2, %2, %4\n\tv_pk_mul_f32 %3, %3, %4\n\t")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
  sink[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[1] + a2[0] + a3[1];
}
// MFMA 16x16x32 f16 back to back (4 independent accumulators), and MFMA interleaved 1:4 with v_exp (does VALU hide in the MFMA shadow?)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NEXP>
__global__ __launch_bounds__(512) void k_mfma(int iters, unsigned long long* out, float* sink) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * threadIdx.x); b[i] = (_Float16)1.0f; }
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float e0 = threadIdx.x * 0.001f, e1 = e0 + 1, e2 = e0 + 2, e3 = e0 + 3;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[j], 0, 0, 0);
        if (NEXP >= 1) asm volatile("v_exp_f32 %0, %0" : "+v"(e0));
        if (NEXP >= 2) asm volatile("v_exp_f32 %0, %0" : "+v"(e1));
        if (NEXP >= 3) asm volatile("v_add_f32 %0, %0, %0" : "+v"(e2));
        if (NEXP >= 4) asm volatile("v_add_f32 %0, %0, %0" : "+v"(e3));
      }
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
  sink[blockIdx.x * 512 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + e0 + e1 + e2 + e3;
}

template <class K>
void run(const char* name, K kern, int per_iter, unsigned long long* d_out, float* sink) {
  const int iters = 2000;
  for (int threads : {64, 256, 512}) {
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, iters, d_out, sink);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, iters, d_out, sink);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 8);
    hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> v;
    for (int b = 0; b < 256; ++b)
      for (int w = 0; w < threads / 64; ++w) v.push_back((double)h[b * 8 + w] / ((double)iters * per_iter));
    std::sort(v.begin(), v.end());
    printf("%-14s waves/SIMD %s: %6.2f cycles per wave-instruction (median; min %.2f max %.2f)  -> %5.2f per SIMD-instruction slot\n", name,
           threads == 64 ? "1/4" : threads == 256 ? "1  " : "2  ", v[v.size() / 2], v.front(), v.back(),
           v[v.size() / 2] / (threads == 512 ? 2.0 : 1.0));
  }
}

int main() {
  unsigned long long* d_out;
  float* sink;
  hipMalloc(&d_out, 256 * 8 * 8);
  hipMalloc(&sink, 256 * 512 * 4);
  run("v_add_f32", k_add, 64, d_out, sink);
  run("v_sub_f32", k_sub, 64, d_out, sink);
  run("v_fma_f32", k_fma, 64, d_out, sink);
  run("v_max_f32", k_max, 64, d_out, sink);
  run("v_max3_f32", k_max3, 64, d_out, sink);
  run("v_maximum3", k_maximum3, 64, d_out, sink);
  run("v_exp_f32", k_exp, 64, d_out, sink);
  run("v_exp_f16", k_exph, 64, d_out, sink);
  run("v_rcp_f32", k_rcp, 64, d_out, sink);
  run("v_cvt_pk_f16", k_cvtpk, 64, d_out, sink);
  run("v_pk_fma_f16", k_pkfmah, 64, d_out, sink);
  run("v_pk_max_f16", k_pkmaxh, 64, d_out, sink);
  run("v_dot2_f32_f16", k_dot2, 64, d_out, sink);
  run("v_dot2c_f32_f16", k_dot2c, 64, d_out, sink);
  run("v_mov_b32", k_mov, 64, d_out, sink);
  run("v_pk_add_f32", k_pkadd, 64, d_out, sink);
  run("v_pk_mul_f32", k_pkmul, 64, d_out, sink);
  run("mfma16x16x32", k_mfma<0>, 16, d_out, sink);
  run("mfma+1exp", k_mfma<1>, 16, d_out, sink);
  run("mfma+2exp", k_mfma<2>, 16, d_out, sink);
  run("mfma+2exp+2add", k_mfma<4>, 16, d_out, sink);
  printf("%s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
