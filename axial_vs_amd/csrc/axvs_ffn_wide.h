// The layer's FFN tail for MANY rows (C = 256): 128 token rows per workgroup.
//
// ffn_fused_kernel streams all of W1 and W2 (1 MiB at F = 1024) through the CU of every 64-row tile, and that private L2 -> register
// stream (~ 80 GB/s per CU, 27.6 k cycles) is longer than the tile's MFMA work (16 k cycles): with more row tiles than CUs the
// kernel pays the stream once per round of the chip (21504 rows = 336 tiles = 2 rounds = 51 us at BASELINE config 3's deformable
// stage).  Here a workgroup owns TWO 64-row halves and every weight fragment it loads multiplies both: half the bytes per row, the
// MFMA work per fragment doubles (MFMA-bound), and 21504 rows are 168 workgroups = one round.
//
// Registers: ONE fragment set (64 VGPRs) that alternates between linear1 and linear2 of a 256-unit chunk -- a slot is refilled
// with the next phase's fragment right behind the MFMAs of its k-step in the second half (in-place refill; the load has the rest
// of that half, the activation epilogue, two barriers and its own k-steps of the next phase to arrive) -- plus the linear2
// accumulators of both halves (64).  LDS: y and h tiles for 128 rows (2 x 64 KiB), no fp32 row tile: norm1 is recomputed from the
// input rows in the epilogue (the same expressions on the same lanes: ffn_finish_kernel does the same).
//
// Bit-exactness: per output element the same MFMA sequence (k-blocks in order, accumulators from zero per chunk, chunks added in
// order) and the same row-wise expressions as ffn_body -- which kernel runs depends on the row count only and must not change a bit.
#pragma once
#include "axvs_fused.h"

namespace axvs {

constexpr int kWideRows = 2 * kRows;
constexpr size_t kFfnWideTiles = 2 * (size_t)kRows * kEpiLd * sizeof(float);      // epilogue: two fp32 row tiles (>= y | h tiles: 128 KiB)
inline size_t ffn_wide_lds_bytes(int F) { return (size_t)(F + 5 * 256) * sizeof(float) + kFfnWideTiles; }

// one half (64 rows) of a GEMM phase with the fragment set `wf`; REFILL (the second half): slot (nt, j) is refilled with the NEXT
// phase's fragment once its MFMAs are issued.  (Both halves k-step by k-step, every slot refilled right behind its use, would
// spread the refills over the whole phase, but the accumulators of both halves -- 64 more VGPRs -- do not fit: 86 spilled.)
template <bool BF, int KB, bool REFILL>
__device__ __forceinline__ void wide_half(f32x4 (&acc)[2][4], u16x8 (&wf)[2][KB], const u16* tile, int fi, int fg,
                                          const u16* __restrict__ Wn, int NRn, int kb0n, int nrow0n) {
  u16x8 bcur[4], bnxt[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) bcur[mt] = act_frag(tile, 0, mt, fi, fg);
#pragma unroll
  for (int j = 0; j < KB; ++j) {
    if (j + 1 < KB) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) bnxt[mt] = act_frag(tile, j + 1, mt, fi, fg);
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = H16<BF>::mfma(wf[nt][j], bcur[mt], acc[nt][mt]);
    if constexpr (REFILL) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) wf[nt][j] = w_frag_u(Wn, NRn, kb0n + j, nrow0n + nt * 16, fg * 16 + fi);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) bcur[mt] = bnxt[mt];
  }
}

template <bool BF, bool GELU = false>
__global__ __launch_bounds__(512) void ffn_wide_kernel(const float* __restrict__ X, const u16* __restrict__ W1, const float* __restrict__ b1,
                                                       const u16* __restrict__ W2, const float* __restrict__ b2,
                                                       const float* __restrict__ g1, const float* __restrict__ be1,
                                                       const float* __restrict__ g2, const float* __restrict__ be2,
                                                       float* __restrict__ out, long long M, int F, RowStride rs) {
  constexpr int C = 256, KB = 8;
  extern __shared__ __attribute__((aligned(16))) char smem_c[];
  FfnLds l;                                            // (xtile unused here)
  l.par = reinterpret_cast<float*>(smem_c);
  char* tiles = smem_c + (size_t)(F + 5 * 256) * sizeof(float);
  u16* ytile[2] = {reinterpret_cast<u16*>(tiles), reinterpret_cast<u16*>(tiles) + KB * kTileElems};
  u16* htile[2] = {ytile[1] + KB * kTileElems, ytile[1] + 2 * KB * kTileElems};
  float* etile[2] = {reinterpret_cast<float*>(tiles), reinterpret_cast<float*>(tiles) + kRows * kEpiLd};
  const int tid = threadIdx.x, lane = tid & 63, fi = lane & 15, fg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform (SGPR): the weight-fragment addresses split into s[base] + lane
  const long long m0 = (long long)blockIdx.x * kWideRows;
  const int nchunk = F / 256;
  const float* sb1 = l.par;
  const float* sb2 = l.par + F;
  const float *sg1 = sb2 + C, *sbe1 = sb2 + 2 * C, *sg2 = sb2 + 3 * C, *sbe2 = sb2 + 4 * C;

  u16x8 wf[2][KB];
  load_wfrags<2, KB>(wf, W1, F, 0, wave * 32, fi, fg, 0);
  ffn_stage_params(l, b1, b2, g1, be1, g2, be2, F, tid);
  // ---- norm1 of my 2 x 8 rows -> y (16-bit) tiles ----
  {
    const float4 gg = *reinterpret_cast<const float4*>(g1 + lane * 4), bb = *reinterpret_cast<const float4*>(be1 + lane * 4);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float4 rows[8];
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        const long long m = min(m0 + h * kRows + wave * 8 + rr, M - 1);
        rows[rr] = *reinterpret_cast<const float4*>(X + rs.row(m) * C + lane * 4);
      }
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        const int r = wave * 8 + rr;
        const float4 v = rows[rr];
        const float mu = wave_sum(v.x + v.y + v.z + v.w) * (1.f / C);
        const float a = v.x - mu, b = v.y - mu, c = v.z - mu, d = v.w - mu;
        const float rstd = rsqrtf(wave_sum(a * a + b * b + c * c + d * d) * (1.f / C) + 1e-5f);
        const f32x4 y = {a * rstd * gg.x + bb.x, b * rstd * gg.y + bb.y, c * rstd * gg.z + bb.z, d * rstd * gg.w + bb.w};
        act_store4<BF>(ytile[h], lane * 4, r, y);
        if (rr == 3) lds_fence();
      }
    }
  }
  __syncthreads();                                     // y tiles and parameters staged

  f32x4 acc2[2][2][4];
  for (int ci = 0; ci < nchunk; ++ci) {
    const int cn = min(ci + 1, nchunk - 1);
    if (ci > 0) __syncthreads();                       // every wave is done reading the previous chunk's h
    // ---- linear1 + activation: my 32 hidden units of the chunk, half by half; in the second half the slots are refilled with the
    //      chunk's linear2 fragments ----
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x4 acc1[2][4];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc1[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (h == 0) wide_half<BF, KB, false>(acc1, wf, ytile[0], fi, fg, W2, C, ci * 8, wave * 32);
      else wide_half<BF, KB, true>(acc1, wf, ytile[1], fi, fg, W2, C, ci * 8, wave * 32);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int hn = ci * 256 + wave * 32 + nt * 16 + fg * 4;
        const float4 bias = *reinterpret_cast<const float4*>(sb1 + hn);
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
          act_store4<BF>(htile[h], wave * 32 + nt * 16 + fg * 4, mt * 16 + fi, v);
        }
      }
    }
    __syncthreads();
    // ---- linear2 partial of the chunk, from zero, then added to the running sum (the order ffn_body keeps); second half: slots
    //      refilled with the next chunk's linear1 fragments (the last chunk re-loads its own: branch-free vmcnt bookkeeping) ----
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x4 part[2][4];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) part[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (h == 0) wide_half<BF, KB, false>(part, wf, htile[0], fi, fg, W1, F, 0, cn * 256 + wave * 32);
      else wide_half<BF, KB, true>(part, wf, htile[1], fi, fg, W1, F, 0, cn * 256 + wave * 32);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc2[h][a][b] = ci == 0 ? part[a][b] : acc2[h][a][b] + part[a][b];
    }
  }
  __syncthreads();                                     // the h tiles are free: the fp32 row tiles overlay y | h

  // ---- accumulators -> fp32 rows in LDS; then per whole row: y = norm1(x) again, + acc2 + b2, norm2, one 1-KiB store ----
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) epi_put(etile[h], mt * 16 + fi, wave * 32 + nt * 16 + fg * 4, acc2[h][nt][mt]);
    lds_fence();
  }
  __syncthreads();
  {
    const float4 gg = *reinterpret_cast<const float4*>(sg1 + lane * 4), bb = *reinterpret_cast<const float4*>(sbe1 + lane * 4);
    const float4 bv = *reinterpret_cast<const float4*>(sb2 + lane * 4);
    const float4 g2v = *reinterpret_cast<const float4*>(sg2 + lane * 4), be2v = *reinterpret_cast<const float4*>(sbe2 + lane * 4);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float4 xr[8];
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        const long long m = min(m0 + h * kRows + wave * 8 + rr, M - 1);
        xr[rr] = *reinterpret_cast<const float4*>(X + rs.row(m) * C + lane * 4);
      }
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        const int r = wave * 8 + rr;
        const float4 v = xr[rr];
        const float mu = wave_sum(v.x + v.y + v.z + v.w) * (1.f / C);
        const float a = v.x - mu, b = v.y - mu, c = v.z - mu, d = v.w - mu;
        const float rstd = rsqrtf(wave_sum(a * a + b * b + c * c + d * d) * (1.f / C) + 1e-5f);
        const float4 y = float4{a * rstd * gg.x + bb.x, b * rstd * gg.y + bb.y, c * rstd * gg.z + bb.z, d * rstd * gg.w + bb.w};
        const float4 ac = *reinterpret_cast<const float4*>(etile[h] + r * kEpiLd + lane * 4);
        const float4 u = float4{y.x + ac.x + bv.x, y.y + ac.y + bv.y, y.z + ac.z + bv.z, y.w + ac.w + bv.w};
        const float mu2 = wave_sum(u.x + u.y + u.z + u.w) * (1.f / C);
        const float d0 = u.x - mu2, d1 = u.y - mu2, d2 = u.z - mu2, d3 = u.w - mu2;
        const float rstd2 = rsqrtf(wave_sum(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3) * (1.f / C) + 1e-5f);
        const long long m = m0 + h * kRows + r;
        if (m < M)
          *reinterpret_cast<float4*>(out + rs.row(m) * C + lane * 4) =
              float4{d0 * rstd2 * g2v.x + be2v.x, d1 * rstd2 * g2v.y + be2v.y, d2 * rstd2 * g2v.z + be2v.z, d3 * rstd2 * g2v.w + be2v.w};
        if (rr == 3) lds_fence();
      }
    }
  }
}

}  // namespace axvs
