// Probe: WAR between an MFMA's source registers and an LDS read that overwrites them right after the MFMA is issued,
// with and without an MFMA-dense partner wave on the same SIMD (waves w and w+4 of a 512-thread workgroup).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void probe(int iters, int partner, int nmfma, unsigned* bad, float* sink) {
  __shared__ __attribute__((aligned(16))) _Float16 lds[2][64 * 8];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (threadIdx.x < 64) for (int i = 0; i < 8; ++i) { lds[0][lane * 8 + i] = (_Float16)1.0f; lds[1][lane * 8 + i] = (_Float16)3.0f; }
  __syncthreads();
  f16x8 a;
  for (int i = 0; i < 8; ++i) a[i] = (_Float16)1.0f;
  if (wave >= 4) {
    if (!partner) return;
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int it = 0; it < iters * 16; ++it) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, a, acc[j], 0, 0, 0);
    }
    sink[blockIdx.x * 512 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    return;
  }
  unsigned nbad = 0;
  const unsigned addr0 = (unsigned)(size_t)(&lds[0][lane * 8]) , addr1 = (unsigned)(size_t)(&lds[1][lane * 8]);
  for (int it = 0; it < iters; ++it) {
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0};
    u32x4 b;
    // b <- ones ; MFMA x4 reading b ; b <- threes (overwrite right after the MFMAs issue) ; results must be 32 each
    asm volatile(
        "ds_read_b128 %4, %5\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_mfma_f32_16x16x32_f16 %0, %6, %4, %0\n\t"
        "v_mfma_f32_16x16x32_f16 %1, %6, %4, %1\n\t"
        "v_mfma_f32_16x16x32_f16 %2, %6, %4, %2\n\t"
        "v_mfma_f32_16x16x32_f16 %3, %6, %4, %3\n\t"
        "ds_read_b128 %4, %7\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_nop 15\n\ts_nop 15\n\t"
        : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3), "=&v"(b)
        : "v"(addr0), "v"(a), "v"(addr1)
        : "memory");
    if (acc0[0] != 32.f || acc1[1] != 32.f || acc2[2] != 32.f || acc3[3] != 32.f) ++nbad;
    if (b[0] != 0x42004200u) ++nbad;   // 3.0h,3.0h
  }
  if (nbad) atomicAdd(bad, nbad);
}

int main() {
  unsigned* bad; float* sink;
  (void)hipMalloc(&bad, 4); (void)hipMalloc(&sink, 1024 * 512 * 4);
  for (int partner = 0; partner < 2; ++partner) {
    (void)hipMemset(bad, 0, 4);
    hipLaunchKernelGGL(probe, dim3(1024), dim3(512), 0, 0, 20000, partner, 4, bad, sink);
    (void)hipDeviceSynchronize();
    unsigned h; (void)hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
    printf("partner=%d mismatches=%u  (%s)\n", partner, h, hipGetErrorString(hipGetLastError()));
  }
  return 0;
}
