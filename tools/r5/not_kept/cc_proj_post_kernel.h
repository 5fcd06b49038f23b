// Cross-clip layer chain, round 5: the 1x1 projection of the temporal ASPP and the block's tail in ONE launch.
//   CC/maxtron_cross_clip_tracking_module.py:186-201 (ASPP: concat of the three dilated branches -> ConvBN(768 -> 256, bias = False,
//   norm = channels-first LayerNorm eps 1e-6, act = GELU)) and :293-295 (conv_norms[i]((aspp(x) + x)^T): LayerNorm eps 1e-5).
// Until round 4 this was a 64 x 64-tile GEMM writing y fp32 (4.7 us) followed by the row-wise cc_aspp_post_kernel (4.6 us): two
// dependent launches of a chain that is launch-latency-bound (512 rows).  Here a workgroup owns 16 rows: the 768-wide concatenated
// branch outputs (blocked 16-bit, written by the batched branch GEMM) are staged in LDS (24 KiB), the 8 waves split the 256 output
// channels (N-split: every weight fragment goes L2 -> VGPR to exactly one wave, a rolling window of 8 k-steps), the accumulators
// pass through an fp32 LDS tile and every wave finishes two whole rows: y -> LN(eps 1e-6) -> GELU -> + x -> LN(eps 1e-5) -> out.
#pragma once
#include "axvs_common.h"
#include "axvs_fused.h"

namespace axvs {

template <bool BF>
__global__ __launch_bounds__(512) void cc_proj_post_kernel(const u16* __restrict__ cat16 /* [24][R][32] blocked 16-bit */,
                                                           const u16* __restrict__ Wp /* packed [256 rows][768] */,
                                                           const float* __restrict__ Xin, const float* __restrict__ ga,
                                                           const float* __restrict__ ba, const float* __restrict__ gn,
                                                           const float* __restrict__ bn, float* __restrict__ out,
                                                           float* __restrict__ out2 /* nullable second copy */, long long R) {
  constexpr int C = 256, KB = 24, ROWS = 16;
  __shared__ __attribute__((aligned(16))) u16 atile[KB * ROWS * 32];
  __shared__ __attribute__((aligned(16))) float etile[ROWS * kEpiLd];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fi = lane & 15, fg = lane >> 4;
  const long long m0 = (long long)blockIdx.x * ROWS;
  // my 32 output channels: the first 8 k-steps of fragments, requested before the tile is staged
  u16x8 wf[2][8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) wf[nt][j] = w_frag(Wp, C, j, wave * 32 + nt * 16 + fi, fg);
  // stage the 16 x 768 operand tile: 1536 chunks of 16 bytes, 3 per thread (rows past the end are clamped copies, never stored)
  {
    u16x8 v[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const int c = tid + p * 512, g = c & 3, row = (c >> 2) & 15, kb = c >> 6;
      const long long m = min(m0 + row, R - 1);
      v[p] = *reinterpret_cast<const u16x8*>(cat16 + ((long long)kb * R + m) * 32 + g * 8);
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const int c = tid + p * 512, g = c & 3, row = (c >> 2) & 15, kb = c >> 6;
      *reinterpret_cast<u16x8*>(atile + (kb * ROWS + row) * 32 + swz_chunk(row, g) * 8) = v[p];
    }
  }
  __syncthreads();
  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  const int boff = fi * 32 + swz_chunk(fi, fg) * 8;
#pragma unroll
  for (int jb = 0; jb < 3; ++jb) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int kb = jb * 8 + j;
      const u16x8 b = *reinterpret_cast<const u16x8*>(atile + kb * ROWS * 32 + boff);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        acc[nt] = H16<BF>::mfma(wf[nt][j], b, acc[nt]);                       // D[channel][row]
        if (jb < 2) wf[nt][j] = w_frag(Wp, C, kb + 8, wave * 32 + nt * 16 + fi, fg);      // the same slot, 8 k-steps on
      }
    }
  }
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) epi_put(etile, fi, wave * 32 + nt * 16 + fg * 4, acc[nt]);
  __syncthreads();
  // ---- two whole rows per wave: z = GELU(LN_cf(y; 1e-6) * g_a + b_a) + x;  out = LN(z; 1e-5) * g_n + b_n ----
  const float4 g = *reinterpret_cast<const float4*>(ga + lane * 4), b = *reinterpret_cast<const float4*>(ba + lane * 4);
  const float4 g2 = *reinterpret_cast<const float4*>(gn + lane * 4), b2 = *reinterpret_cast<const float4*>(bn + lane * 4);
  auto gelu = [](float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752f)); };
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = wave * 2 + i;
    const long long m = m0 + row;
    if (m >= R) continue;
    const float4 y = *reinterpret_cast<const float4*>(etile + row * kEpiLd + lane * 4);
    const float4 x = *reinterpret_cast<const float4*>(Xin + m * C + lane * 4);
    const float mu = wave_sum(y.x + y.y + y.z + y.w) * (1.f / C);
    float d0 = y.x - mu, d1 = y.y - mu, d2 = y.z - mu, d3 = y.w - mu;
    const float rstd = 1.f / sqrtf(wave_sum(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3) * (1.f / C) + 1e-6f);
    const float z0 = gelu(d0 * rstd * g.x + b.x) + x.x, z1 = gelu(d1 * rstd * g.y + b.y) + x.y;
    const float z2 = gelu(d2 * rstd * g.z + b.z) + x.z, z3 = gelu(d3 * rstd * g.w + b.w) + x.w;
    const float mu2 = wave_sum(z0 + z1 + z2 + z3) * (1.f / C);
    d0 = z0 - mu2; d1 = z1 - mu2; d2 = z2 - mu2; d3 = z3 - mu2;
    const float rstd2 = rsqrtf(wave_sum(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3) * (1.f / C) + 1e-5f);
    const float4 o = float4{d0 * rstd2 * g2.x + b2.x, d1 * rstd2 * g2.y + b2.y, d2 * rstd2 * g2.z + b2.z, d3 * rstd2 * g2.w + b2.w};
    *reinterpret_cast<float4*>(out + m * C + lane * 4) = o;
    if (out2) *reinterpret_cast<float4*>(out2 + m * C + lane * 4) = o;
  }
}

}  // namespace axvs
