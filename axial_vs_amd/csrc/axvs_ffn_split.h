// The layer's FFN tail for FEW rows (C = 256): the hidden units are split over workgroups.
//
// ffn_fused_kernel gives every 64-row tile one workgroup that streams all of W1 and W2 (2 F C 16-bit words = 1 MiB at F = 1024)
// through its own CU; with a few dozen row tiles (the coarse pyramid levels of the pixel decoder: 1024 .. 4096 tokens) that private
// stream is the whole run time (25 us at 71 GB/s per CU) while most of the chip idles.  Here workgroup (tile, chunk) computes one
// 256-unit chunk of the hidden layer for its 64 rows -- 256 KiB of weights -- and writes the partial linear2 output; a second,
// row-wise kernel adds the partials IN CHUNK ORDER, the residual and the bias and applies norm2.
//
// Bit-exactness: ffn_body (axvs_fused.h) accumulates every chunk's linear2 contribution from zero and adds the chunks in order, and
// the row-wise pieces below are the same expressions on the same lane <-> channel mapping, so the two forms agree to the last
// bit -- which form runs depends on the number of rows, and a clip's result must not depend on its batch (batch sharding).
#pragma once
#include "axvs_fused.h"

namespace axvs {

constexpr size_t kFfnSplitLds = 2 * 8 * kTileElems * sizeof(u16) + (size_t)kRows * kEpiLd * sizeof(float);   // y | h | fp32 rows

// CPW: chunks per workgroup.  1: one 256-unit chunk (up to 64 tiles: tiles x F/256 workgroups fit one round of the chip at one workgroup per CU);
// 2: two consecutive chunks, one after the other (65 .. 128 tiles: tiles x F/512 workgroups still fit one round; the next chunk's linear1 fragments are
// requested behind the linear2 products of the current one).  Every chunk's partial is accumulated from zero and written on its own: same bits.
template <bool BF, bool GELU = false, int CPW = 1>
__global__ __launch_bounds__(512) void ffn_split_kernel(const float* __restrict__ X, const u16* __restrict__ W1, const float* __restrict__ b1,
                                                        const u16* __restrict__ W2, const float* __restrict__ g1,
                                                        const float* __restrict__ be1, float* __restrict__ part /* [F/256][M][256] */,
                                                        long long M, int F, RowStride rs) {
  constexpr int C = 256, KB = 8;
  extern __shared__ __attribute__((aligned(16))) char smem_c[];
  u16* ytile = reinterpret_cast<u16*>(smem_c);
  u16* htile = ytile + KB * kTileElems;
  float* etile = reinterpret_cast<float*>(htile + KB * kTileElems);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fi = lane & 15, fg = lane >> 4;
  const long long m0 = (long long)blockIdx.x * kRows;
  const int c0 = blockIdx.y * CPW;                            // first hidden-unit chunk of this workgroup
  u16x8 w1f[2][KB], w2f[2][KB];
  load_wfrags<2, KB>(w1f, W1, F, 0, c0 * 256 + wave * 32, fi, fg, 0);
  // ---- norm1 of my 8 rows -> y (16-bit) tile; the same expressions as ffn_body ----
  {
    const float4 gg = *reinterpret_cast<const float4*>(g1 + lane * 4), bb = *reinterpret_cast<const float4*>(be1 + lane * 4);
    float4 rows[8];
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const long long m = min(m0 + wave * 8 + rr, M - 1);
      rows[rr] = *reinterpret_cast<const float4*>(X + rs.row(m) * C + lane * 4);
    }
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int r = wave * 8 + rr;
      const float4 v = rows[rr];
      const float mu = wave_sum(v.x + v.y + v.z + v.w) * (1.f / C);
      const float a = v.x - mu, b = v.y - mu, cc = v.z - mu, d = v.w - mu;
      const float rstd = rsqrtf(wave_sum(a * a + b * b + cc * cc + d * d) * (1.f / C) + 1e-5f);
      const f32x4 y = {a * rstd * gg.x + bb.x, b * rstd * gg.y + bb.y, cc * rstd * gg.z + bb.z, d * rstd * gg.w + bb.w};
      act_store4<BF>(ytile, lane * 4, r, y);
      if (rr == 3) lds_fence();
    }
  }
  __syncthreads();
#pragma unroll
  for (int cc = 0; cc < CPW; ++cc) {
  const int c = c0 + cc;
  // ---- linear1 + ReLU for the chunk (meanwhile fetch the chunk's linear2 fragments) ----
  f32x4 acc1[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc1[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  gemm_phase_pf<BF, 2, 4, KB, 2>(acc1, w1f, ytile, fi, fg, 0, w2f, W2, C, c * 8, wave * 32, 0);
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int hn = c * 256 + wave * 32 + nt * 16 + fg * 4;
    const float4 bias = *reinterpret_cast<const float4*>(b1 + hn);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      f32x4 v = acc1[nt][mt];
      if constexpr (GELU) {
        v[0] = gelu_exact(v[0] + bias.x); v[1] = gelu_exact(v[1] + bias.y);
        v[2] = gelu_exact(v[2] + bias.z); v[3] = gelu_exact(v[3] + bias.w);
      } else {
        v[0] = fmaxf(v[0] + bias.x, 0.f); v[1] = fmaxf(v[1] + bias.y, 0.f);
        v[2] = fmaxf(v[2] + bias.z, 0.f); v[3] = fmaxf(v[3] + bias.w, 0.f);
      }
      act_store4<BF>(htile, wave * 32 + nt * 16 + fg * 4, mt * 16 + fi, v);
    }
  }
  __syncthreads();
  // ---- linear2 partial of the chunk, accumulated from zero ----
  f32x4 p2[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) p2[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (cc + 1 < CPW) gemm_phase_pf<BF, 2, 4, KB, 2>(p2, w2f, htile, fi, fg, 0, w1f, W1, F, 0, (c + 1) * 256 + wave * 32, 0);
  else gemm_phase<BF, 2, 4, KB>(p2, w2f, htile, fi, fg, 0);
  // accumulator layout -> fp32 rows in LDS -> one 1-KiB store per row
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) epi_put(etile, mt * 16 + fi, wave * 32 + nt * 16 + fg * 4, p2[nt][mt]);
  __syncthreads();
  float* dst = part + (long long)c * M * C;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = wave * 8 + i;
    if (m0 + r < M) *reinterpret_cast<float4*>(dst + (m0 + r) * C + lane * 4) = *reinterpret_cast<const float4*>(etile + r * kEpiLd + lane * 4);
  }
  if (cc + 1 < CPW) __syncthreads();        // the fp32 rows are read: the next chunk may write them (its h tile is fenced by the barriers above)
  }
}

// out = norm2(y + (p_0 + p_1 + ...) + b2),  y = norm1(x): one wave per row, lane = 4 channels (as in ffn_body)
__global__ __launch_bounds__(256) void ffn_finish_kernel(const float* __restrict__ X, const float* __restrict__ part, const float* __restrict__ b2,
                                                         const float* __restrict__ g1, const float* __restrict__ be1,
                                                         const float* __restrict__ g2, const float* __restrict__ be2,
                                                         float* __restrict__ out, long long M, int nchunk, RowStride rs) {
  constexpr int C = 256;
  const long long m = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (m >= M) return;
  const long long mrow = rs.row(m);                     // X and out rows (the partial sums are in natural order)
  const float4 gg = *reinterpret_cast<const float4*>(g1 + lane * 4), bb = *reinterpret_cast<const float4*>(be1 + lane * 4);
  const float4 v = *reinterpret_cast<const float4*>(X + mrow * C + lane * 4);
  const float mu = wave_sum(v.x + v.y + v.z + v.w) * (1.f / C);
  const float a = v.x - mu, b = v.y - mu, cc = v.z - mu, d = v.w - mu;
  const float rstd = rsqrtf(wave_sum(a * a + b * b + cc * cc + d * d) * (1.f / C) + 1e-5f);
  const float4 y = float4{a * rstd * gg.x + bb.x, b * rstd * gg.y + bb.y, cc * rstd * gg.z + bb.z, d * rstd * gg.w + bb.w};
  float4 acc = *reinterpret_cast<const float4*>(part + m * C + lane * 4);
  for (int c = 1; c < nchunk; ++c) {
    const float4 p = *reinterpret_cast<const float4*>(part + ((long long)c * M + m) * C + lane * 4);
    acc.x = acc.x + p.x; acc.y = acc.y + p.y; acc.z = acc.z + p.z; acc.w = acc.w + p.w;
  }
  const float4 bv = *reinterpret_cast<const float4*>(b2 + lane * 4);
  const float4 u = float4{y.x + acc.x + bv.x, y.y + acc.y + bv.y, y.z + acc.z + bv.z, y.w + acc.w + bv.w};
  const float mu2 = wave_sum(u.x + u.y + u.z + u.w) * (1.f / C);
  const float d0 = u.x - mu2, d1 = u.y - mu2, d2 = u.z - mu2, d3 = u.w - mu2;
  const float rstd2 = rsqrtf(wave_sum(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3) * (1.f / C) + 1e-5f);
  const float4 g2v = *reinterpret_cast<const float4*>(g2 + lane * 4), be2v = *reinterpret_cast<const float4*>(be2 + lane * 4);
  *reinterpret_cast<float4*>(out + mrow * C + lane * 4) =
      float4{d0 * rstd2 * g2v.x + be2v.x, d1 * rstd2 * g2v.y + be2v.y, d2 * rstd2 * g2v.z + be2v.z, d3 * rstd2 * g2v.w + be2v.w};
}

}  // namespace axvs
