// fp32 GEMMs of the training tier's Linear layers on the 16-bit matrix cores, in split precision ("bf16x3").
//
// Every fp32 operand x is split into two bf16 numbers, x = hi + lo + O(2^-17 |x|) (hi = bf16(x), lo = bf16(x - hi)), while its
// tile is staged into LDS, and a product is three MFMAs, hi.hi + hi.lo + lo.hi, accumulated in fp32: 16 mantissa bits per
// operand (relative error of a product ~1.5e-5; lo.lo, ~2^-16 of the product, is dropped with the split's own residual), the
// fp32 exponent range (gradients of 1e-8 are ordinary numbers -- an fp16 split would need loss scaling), and 3/16 of the matrix
// pipe time of the fp32 MFMA (v_mfma_f32_16x16x4_f32 runs at 1/16 of the bf16 rate).  This replaces the rocBLAS sgemm calls of
// round 2 (fp32 MFMA inside Tensile: ~110 TFLOP/s here); the training fixtures hold the tier to 1e-4, these kernels sit at ~1e-5.
//
//   tr_gemm_nt_kernel   C[M,N] = epilogue(A[M,K] B[N,K]^T)     forward Linear (x W^T) and, through a transposed copy of the weight,
//                                                            the input gradients (dY W); epilogue: + bias, * mul, ReLU, dropout by
//                                                            element index, + beta C  -- what used to be separate elementwise launches
//   tr_gemm_tn_kernel   P[s][N,K] = sum_{m in split s} dY[m,N] X[m,K]     weight gradients: the contraction runs over the rows, so both
//                                                            operand tiles sit in LDS contraction-major and reach the MFMA through the
//                                                            transposing LDS load (ds_read_b64_tr_b16); partial sums per row split are
//                                                            added by a second, deterministic kernel (no atomics)
// Block tile 128 x 128, k-step 32, 8 waves (wave tile 64 x 32: 24 MFMAs per k-step), operand tiles double-buffered in LDS with the
// global loads of step k+1 in flight under the MFMAs of step k, one barrier per step; two workgroups per CU (<= 128 VGPRs).
#pragma once
#include <type_traits>

#include "axvs_train.h"

namespace axvs {
namespace tr {

__device__ __forceinline__ void split8(const float4& a, const float4& b, u16x8& hi, u16x8& lo) {
  const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 h = (__bf16)x[i];
    hi[i] = __builtin_bit_cast(u16, h);
    lo[i] = __builtin_bit_cast(u16, (__bf16)(x[i] - (float)h));
  }
}

__device__ __forceinline__ f32x4 mfma3(u16x8 ah, u16x8 al, u16x8 bh, u16x8 bl, f32x4 c) {   // c += (ah + al) . (bh + bl), lo.lo dropped
  c = H16<true>::mfma(ah, bh, c);
  c = H16<true>::mfma(ah, bl, c);
  return H16<true>::mfma(al, bh, c);
}

// ---- P[split][N,K] = sum over the split's rows m of dY[m,N] X[m,K]: both operands contraction-major -----------------------------------
// LDS image of an operand tile: [32 rows m][128 columns] 16-bit, 256-byte rows, 16-byte chunk c of row r at chunk c ^ f(r),
// f(r) = ((r & 3) << 2) | ((r >> 2) & 3): the staging stores (16 lanes per row) and the transposed fragment reads (per 32-lane
// half: 8 rows x one 32-byte column pair) both hit 8 distinct 32-byte slots of the 256-byte bank row.
__device__ __forceinline__ int tn_chunk(int r, int c) { return c ^ (((r & 3) << 2) | ((r >> 2) & 3)); }

// transposed read of a 16 (columns) x 32 (rows m) fragment whose columns start at `col0` (multiple of 16): lane (i = lane & 15,
// g = lane >> 4) receives column col0 + i at rows 8 g .. 8 g + 7 -- the k order of the MFMA operands
__device__ __forceinline__ u16x8 tn_frag(const u16* tile, int col0, int fi, int fg) {
  typedef short s16x4v __attribute__((ext_vector_type(4)));
  u16x8 r;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = 8 * fg + 4 * h + (fi >> 2);                  // this lane supplies the address of row `row`, columns col0 + 4 (fi & 3) ..
    const int col = col0 + 4 * (fi & 3);
    const s16x4v t4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (s16x4v __attribute__((address_space(3)))*)(tile + row * kGT + tn_chunk(row, col >> 3) * 8 + (col & 7)));
#pragma unroll
    for (int e = 0; e < 4; ++e) r[4 * h + e] = (u16)t4[e];
  }
  return r;
}

// part_b (nullable): [split][N] partial COLUMN SUMS of dY (the bias gradient of the same Linear layer), added in fp32 from the
// staging registers by the workgroups of the first k-tile (a bias gradient can be a sum that cancels to zero -- the k bias of a
// softmax -- so it does not go through the bf16 split) and reduced over the 32 staging rows through LDS in a fixed order.
// AMP = 1 / 2 (option train_amp, under torch.autocast): ONE bf16 / fp16 piece per operand, as in tr_gemm_nt_kernel<1>.
// STATS: GemmLd::stat_* -- sums of the output tile for the BatchNorm that follows the mask einsum
template <bool GEN, int AMP = 0, bool STATS = false>      // GEN: rows at any 4-byte boundary / extents that are not multiples of 4 (see tr_gemm_nt_kernel)
__global__ __launch_bounds__(512, 4) void tr_gemm_tn_kernel(const float* __restrict__ dY, const float* __restrict__ X, float* __restrict__ part,
                                                            long long M, int N, int K, long long rows_per_split,
                                                            float* __restrict__ part_b, GemmLd ld) {
  extern __shared__ __attribute__((aligned(16))) char gsmem[];
  u16* const sbuf = reinterpret_cast<u16*>(gsmem);            // [2 stages][dY hi | dY lo | X hi | X lo][32 rows m][128]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fi = lane & 15, fg = lane >> 4;
  const int wn = wave >> 2, wk = wave & 3;                   // wave tile: 64 (n) x 32 (k)
  const int tiles_k = (K + kGT - 1) / kGT;
  const int n0 = (blockIdx.x / tiles_k) * kGT, k0 = (blockIdx.x % tiles_k) * kGT;
  const long long ms = (long long)blockIdx.y * rows_per_split;
  const long long me = ms + rows_per_split < M ? ms + rows_per_split : M;
  // staging: thread -> (row m of the step, 8-float chunk): 16 threads cover a 512-byte row segment
  const int sr = tid >> 4, sc = tid & 15;
  const int soff = sr * kGT + tn_chunk(sr, sc) * 8;
  const int nsteps = (int)((me - ms + kGK - 1) / kGK);
  const bool do_bias = part_b != nullptr && k0 == 0;                 // workgroup-uniform
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};         // my 8 columns of dY, rows m = sr (mod 32) of the split
  float4 ra[2], rb[2];
  auto gload = [&](int s) {
    const long long m = ms + (long long)s * kGK + sr;
    const bool m_ok = m < me;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if constexpr (!GEN) {              // 16-byte rows, extents in whole groups of four
        ra[h] = m_ok && n0 + sc * 8 + 4 * h < N ? *reinterpret_cast<const float4*>(dY + m * ld.a + n0 + sc * 8 + 4 * h) : float4{0.f, 0.f, 0.f, 0.f};
        rb[h] = m_ok && k0 + sc * 8 + 4 * h < K ? *reinterpret_cast<const float4*>(X + m * ld.b + k0 + sc * 8 + 4 * h) : float4{0.f, 0.f, 0.f, 0.f};
      } else {
        ra[h] = ldg4(dY + m * ld.a + n0 + sc * 8 + 4 * h, m_ok ? N - (n0 + sc * 8 + 4 * h) : 0, ld.al_a);
        rb[h] = ldg4(X + m * ld.b + k0 + sc * 8 + 4 * h, m_ok ? K - (k0 + sc * 8 + 4 * h) : 0, ld.al_b);
      }
    }
  };
  auto lstore = [&](int stage) {
    u16* const base = sbuf + stage * 4 * kGTileElems;
    u16x8 hi, lo;
    if (do_bias) {
      bsum[0] += ra[0].x; bsum[1] += ra[0].y; bsum[2] += ra[0].z; bsum[3] += ra[0].w;
      bsum[4] += ra[1].x; bsum[5] += ra[1].y; bsum[6] += ra[1].z; bsum[7] += ra[1].w;
    }
    if constexpr (AMP != 0) {
      u16x8 one[1];
      split_n<1, AMP == 2>(ra[0], ra[1], one);
      *reinterpret_cast<u16x8*>(base + soff) = one[0];
      split_n<1, AMP == 2>(rb[0], rb[1], one);
      *reinterpret_cast<u16x8*>(base + 2 * kGTileElems + soff) = one[0];
    } else {
      split8(ra[0], ra[1], hi, lo);
      *reinterpret_cast<u16x8*>(base + soff) = hi;
      *reinterpret_cast<u16x8*>(base + kGTileElems + soff) = lo;
      split8(rb[0], rb[1], hi, lo);
      *reinterpret_cast<u16x8*>(base + 2 * kGTileElems + soff) = hi;
      *reinterpret_cast<u16x8*>(base + 3 * kGTileElems + soff) = lo;
    }
  };
  f32x4 acc[4][2];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) acc[nt][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (nsteps > 0) {
    gload(0);
    lstore(0);
  }
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    const int cur = s & 1;
    if (s + 1 < nsteps) gload(s + 1);
    const u16* const base = sbuf + cur * 4 * kGTileElems;
    u16x8 xh[2], xl[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      xh[kt] = tn_frag(base + 2 * kGTileElems, wk * 32 + kt * 16, fi, fg);
      if constexpr (AMP == 0) xl[kt] = tn_frag(base + 3 * kGTileElems, wk * 32 + kt * 16, fi, fg);
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const u16x8 yh = tn_frag(base, wn * 64 + nt * 16, fi, fg);
      if constexpr (AMP == 0) {
        const u16x8 yl = tn_frag(base + kGTileElems, wn * 64 + nt * 16, fi, fg);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) acc[nt][kt] = mfma3(xh[kt], xl[kt], yh, yl, acc[nt][kt]);   // D[k = 4 fg + r][n = fi]
      } else {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) acc[nt][kt] = H16<AMP != 2>::mfma(xh[kt], yh, acc[nt][kt]);
      }
    }
    if (s + 1 < nsteps) lstore(cur ^ 1);
    __syncthreads();
  }
  // partial tile -> fp32 staging tile [n][k] in LDS (the loop's last barrier is behind every wave: the operand stages are free) ->
  // whole rows: 32 lanes x 16 bytes = 512 contiguous bytes per row of part[split][n][..] (straight from the accumulators a store
  // instruction covered 64-byte pieces of 16 rows: the einsum form, whose output is the large operand, ran at 1.5 - 2 TB/s)
  float* const out = part + (size_t)blockIdx.y * N * ld.c;     // (ld.c != K only with a single split: the einsum form below)
  float* const stg = reinterpret_cast<float*>(gsmem);
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
      *reinterpret_cast<float4*>(stg + (wn * 64 + nt * 16 + fi) * kGLd + wk * 32 + kt * 16 + 4 * fg) =
          float4{acc[nt][kt][0], acc[nt][kt][1], acc[nt][kt][2], acc[nt][kt][3]};
  __syncthreads();
  float sa = 0.f, sb = 0.f;
  float sshift = 0.f;
  if constexpr (STATS) sshift = ld.stat_shift ? ld.stat_shift[0] : 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int idx = tid + 512 * i, row = idx >> 5, c4 = idx & 31;
    const int n = n0 + row, k = k0 + 4 * c4;
    const float4 o4 = *reinterpret_cast<const float4*>(stg + row * kGLd + 4 * c4);
    if constexpr (!GEN) {
      if (n < N && k < K) *reinterpret_cast<float4*>(out + (size_t)n * ld.c + k) = o4;
    } else if (n < N) {
      stg4(out + (size_t)n * ld.c + k, o4, K - k, ld.al_c);
    }
    if constexpr (STATS) {
      const float ov[4] = {o4.x, o4.y, o4.z, o4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (n < N && k + e < K) {
          const float d = ov[e] - sshift;
          sa += d;
          sb += d * d;
        }
    }
  }
  if constexpr (STATS) {      // thread -> wave (xor tree) -> workgroup (eight values, in order): a fixed summation order
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      sa += __shfl_xor(sa, o, 64);
      sb += __shfl_xor(sb, o, 64);
    }
    __syncthreads();           // every wave is done with the staging tile
    float* const red = reinterpret_cast<float*>(gsmem);
    if (lane == 0) {
      red[wave * 2] = sa;
      red[wave * 2 + 1] = sb;
    }
    __syncthreads();
    if (tid == 0) {
      float ta = 0.f, tb = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) {
        ta += red[w * 2];
        tb += red[w * 2 + 1];
      }
      const int g = n0 / ld.stat_rows, tile = (int)(blockIdx.x % tiles_k) + tiles_k * ((n0 % ld.stat_rows) / kGT);
      float* const sp = ld.stat_part + ((size_t)g * ld.stat_nblk + ld.stat_blk0 + tile) * 2;
      sp[0] = ta;
      sp[1] = tb;
    }
  }
  if (do_bias) __syncthreads();  // the bias reduction below reuses the staging tile
  if (do_bias) {
    float* const red = reinterpret_cast<float*>(gsmem);        // [32 staging rows][128 columns]
    *reinterpret_cast<float4*>(red + sr * kGT + sc * 8) = float4{bsum[0], bsum[1], bsum[2], bsum[3]};
    *reinterpret_cast<float4*>(red + sr * kGT + sc * 8 + 4) = float4{bsum[4], bsum[5], bsum[6], bsum[7]};
    __syncthreads();
    if (tid < kGT && n0 + tid < N) {
      float t = 0.f;
#pragma unroll 8
      for (int r = 0; r < 32; ++r) t += red[r * kGT + tid];
      part_b[(size_t)blockIdx.y * N + n0 + tid] = t;
    }
  }
}

}  // namespace tr
}  // namespace axvs
