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

constexpr int kGT = 128;                 // block tile (both output dimensions)
constexpr int kGK = 32;                  // contraction step
constexpr int kGTileElems = kGT * kGK;   // one 16-bit operand tile
constexpr int kGLd = kGT + 4;            // fp32 row stride of the epilogue staging tile
constexpr size_t kGemmLds = (size_t)kGT * kGLd * sizeof(float);   // 67.6 KB >= 2 stages x 4 tiles x 8 KB

struct GemmLd {     // row strides (floats, multiples of 4) of A, B, C; ksteps > 0: split-K, that many 32-wide k-steps per blockIdx.z
  long long a, b, c;
  int ksteps;
};

struct GemmEpi {
  const float* bias;   // nullable [N]: added first
  float mul;           // then multiplied
  int relu;            // then max(., 0)
  Drop dr;             // then dropout, element index row * N + col (thr = 0: none)
  float beta;          // 0: overwrite C, 1: add to it
};

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

// ---- C[M,N] = epilogue(A[M,K] . B[N,K]^T): both operands row-major with the contraction index contiguous -----------------------------
// NS = 2: the bf16x3 product above.  NS = 3: operands split into THREE bf16 pieces (24 mantissa bits = all of fp32) and six MFMAs per
// product (hh, hm, mh, mm, hl, lh; the dropped terms are 2^-24 of the product): as accurate as an fp32 GEMM at twice the matrix
// time of NS = 2.  The FORWARD GEMMs run this way (option `train_exact`, default 1).  Why: the layer has one discontinuity, the ReLU.
// With 1.5e-5 relative error on the pre-activations about one hidden unit in 1e5 lands on the other side of zero than in an fp32
// forward, and every such flip moves the input gradient of its token by ~1e-2 of the gradient's scale (measured at [1,4,256,32,32],
// d_ffn 1024, two-piece forward: d_src 7e-3 .. 2e-2 in max-norm, relative L2 3e-4, output 5e-6).  The three-piece forward costs
// 0.06 ms of a 3.3 ms step (these GEMMs are bound by their fp32 operand traffic, not by the matrix pipe), so it is the default;
// the backward GEMMs are smooth in their operands and stay two-piece.
template <int NS>
__device__ __forceinline__ void split_n(const float4& a, const float4& b, u16x8 (&p)[NS]) {
  const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float r = x[i];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const __bf16 h = (__bf16)r;
      p[s][i] = __builtin_bit_cast(u16, h);
      r -= (float)h;                       // exact in fp32
    }
  }
}

template <int NS>
constexpr size_t gemm_nt_lds() {
  return (size_t)2 * 2 * NS * kGTileElems * sizeof(u16) > kGemmLds ? (size_t)2 * 2 * NS * kGTileElems * sizeof(u16) : kGemmLds;
}

template <int NS>
__global__ __launch_bounds__(512, NS == 2 ? 4 : 2) void tr_gemm_nt_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                                         float* __restrict__ C, long long M, int N, int K, GemmLd ld, GemmEpi ep) {
  extern __shared__ __attribute__((aligned(16))) char gsmem[];
  u16* const sbuf = reinterpret_cast<u16*>(gsmem);            // [2 stages][A pieces | B pieces][128 rows][32] (rows chunk-swizzled)
  constexpr int kStage = 2 * NS * kGTileElems;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fi = lane & 15, fg = lane >> 4;
  const int wm = wave >> 2, wn = wave & 3;
  const long long m0 = (long long)blockIdx.x * kGT;
  const int n0 = blockIdx.y * kGT;
  // staging: thread -> (tile row, 8-float chunk): 4 threads cover a 128-byte row segment
  const int sr = tid >> 2, sq = tid & 3;
  const bool a_ok = m0 + sr < M, b_ok = n0 + sr < N;
  // split-K (ld.ksteps > 0): workgroup z contracts k-steps [z ksteps, (z + 1) ksteps) into the partial C + z M ldc
  const int nk_all = (K + kGK - 1) / kGK;
  const int ks0 = ld.ksteps > 0 ? (int)blockIdx.z * ld.ksteps : 0;
  const int nk = ld.ksteps > 0 ? (nk_all - ks0 < ld.ksteps ? (nk_all - ks0 > 0 ? nk_all - ks0 : 0) : ld.ksteps) : nk_all;
  const int kbase = ks0 * kGK;
  const float* ap = A + (m0 + (a_ok ? sr : 0)) * ld.a + kbase + sq * 8;
  const float* bp = B + (long long)(n0 + (b_ok ? sr : 0)) * ld.b + kbase + sq * 8;
  C += (size_t)blockIdx.z * M * ld.c;
  const int soff = sr * 32 + swz_chunk(sr, sq) * 8;
  // global loads run TWO k-steps ahead of the MFMAs (two register slots, used alternately): with one step of lookahead a load had a
  // single step's MFMAs (~0.3 us) to cover an L2 / HBM round trip and every step stalled at its LDS store
  float4 ra[2][2], rb[2][2];
  auto gload = [&](int ks, auto slot_tag) {
    constexpr int SL = decltype(slot_tag)::value;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int kk = kbase + ks * kGK + sq * 8 + 4 * h;       // K % 4 == 0: a float4 is inside or outside
      const bool kin = kk < K;
      ra[SL][h] = a_ok && kin ? *reinterpret_cast<const float4*>(ap + ks * kGK + 4 * h) : float4{0.f, 0.f, 0.f, 0.f};
      rb[SL][h] = b_ok && kin ? *reinterpret_cast<const float4*>(bp + ks * kGK + 4 * h) : float4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto lstore = [&](int stage, auto slot_tag) {
    constexpr int SL = decltype(slot_tag)::value;
    u16* const base = sbuf + stage * kStage;
    u16x8 p[NS];
    split_n<NS>(ra[SL][0], ra[SL][1], p);
#pragma unroll
    for (int s = 0; s < NS; ++s) *reinterpret_cast<u16x8*>(base + s * kGTileElems + soff) = p[s];
    split_n<NS>(rb[SL][0], rb[SL][1], p);
#pragma unroll
    for (int s = 0; s < NS; ++s) *reinterpret_cast<u16x8*>(base + (NS + s) * kGTileElems + soff) = p[s];
  };
  int aoff[4], boff[2];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int row = wm * 64 + mt * 16 + fi;
    aoff[mt] = row * 32 + swz_chunk(row, fg) * 8;
  }
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int row = wn * 32 + nt * 16 + fi;
    boff[nt] = row * 32 + swz_chunk(row, fg) * 8;
  }
  f32x4 acc[4][2];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  // one k-step: stage `PAR` of LDS holds step ks; register slot PAR ^ 1 holds step ks + 1 (requested a step ago), slot PAR is free
  auto kstep = [&](int ks, auto par_tag) {
    constexpr int PAR = decltype(par_tag)::value;
    if (ks + 2 < nk) gload(ks + 2, std::integral_constant<int, PAR>{});
    const u16* const base = sbuf + PAR * kStage;
    u16x8 bf[2][NS];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int s = 0; s < NS; ++s) bf[nt][s] = *reinterpret_cast<const u16x8*>(base + (NS + s) * kGTileElems + boff[nt]);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      u16x8 af[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) af[s] = *reinterpret_cast<const u16x8*>(base + s * kGTileElems + aoff[mt]);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {                        // D[n = 4 fg + r][m = fi]
        f32x4 c = acc[mt][nt];
        c = H16<true>::mfma(bf[nt][0], af[0], c);
        c = H16<true>::mfma(bf[nt][0], af[1], c);
        c = H16<true>::mfma(bf[nt][1], af[0], c);
        if constexpr (NS == 3) {
          c = H16<true>::mfma(bf[nt][1], af[1], c);
          c = H16<true>::mfma(bf[nt][0], af[2], c);
          c = H16<true>::mfma(bf[nt][2], af[0], c);
        }
        acc[mt][nt] = c;
      }
    }
    if (ks + 1 < nk) lstore(PAR ^ 1, std::integral_constant<int, PAR ^ 1>{});   // the other stage: every wave left its reads behind the previous barrier
    __syncthreads();
  };
  if (nk > 0) gload(0, S0{});
  if (nk > 1) gload(1, S1{});
  if (nk > 0) lstore(0, S0{});
  __syncthreads();
  for (int ks = 0; ks < nk; ks += 2) {
    kstep(ks, S0{});
    if (ks + 1 < nk) kstep(ks + 1, S1{});
  }
  // ---- epilogue: accumulators -> fp32 staging tile [m][n] -> whole rows, 512 bytes per row segment ----
  float* const stg = reinterpret_cast<float*>(gsmem);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
      *reinterpret_cast<float4*>(stg + (wm * 64 + mt * 16 + fi) * kGLd + wn * 32 + nt * 16 + 4 * fg) =
          float4{acc[mt][nt][0], acc[mt][nt][1], acc[mt][nt][2], acc[mt][nt][3]};
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int idx = tid + 512 * i, row = idx >> 5, c4 = idx & 31;
    const long long gm = m0 + row;
    const int gn = n0 + 4 * c4;
    if (gm < M && gn < N) {
      const float4 v = *reinterpret_cast<const float4*>(stg + row * kGLd + 4 * c4);
      float t[4] = {v.x, v.y, v.z, v.w};
      if (ep.bias) {
        const float4 b = *reinterpret_cast<const float4*>(ep.bias + gn);
        t[0] += b.x; t[1] += b.y; t[2] += b.z; t[3] += b.w;
      }
      float* const cp = C + gm * ld.c + gn;
      const unsigned long long e0 = (unsigned long long)gm * N + gn;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        t[e] *= ep.mul;
        if (ep.relu) t[e] = fmaxf(t[e], 0.f);
        t[e] *= drop_keep(ep.dr, e0 + e);
      }
      if (ep.beta != 0.f) {
        const float4 o = *reinterpret_cast<const float4*>(cp);
        t[0] += o.x; t[1] += o.y; t[2] += o.z; t[3] += o.w;
      }
      *reinterpret_cast<float4*>(cp) = make_float4(t[0], t[1], t[2], t[3]);
    }
  }
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
  const bool n_ok = n0 + sc * 8 < N, k_ok = k0 + sc * 8 < K;         // N, K % 8 == 0 (host check)
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
      ra[h] = m_ok && n_ok ? *reinterpret_cast<const float4*>(dY + m * ld.a + n0 + sc * 8 + 4 * h) : float4{0.f, 0.f, 0.f, 0.f};
      rb[h] = m_ok && k_ok ? *reinterpret_cast<const float4*>(X + m * ld.b + k0 + sc * 8 + 4 * h) : float4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto lstore = [&](int stage) {
    u16* const base = sbuf + stage * 4 * kGTileElems;
    u16x8 hi, lo;
    if (do_bias) {
      bsum[0] += ra[0].x; bsum[1] += ra[0].y; bsum[2] += ra[0].z; bsum[3] += ra[0].w;
      bsum[4] += ra[1].x; bsum[5] += ra[1].y; bsum[6] += ra[1].z; bsum[7] += ra[1].w;
    }
    split8(ra[0], ra[1], hi, lo);
    *reinterpret_cast<u16x8*>(base + soff) = hi;
    *reinterpret_cast<u16x8*>(base + kGTileElems + soff) = lo;
    split8(rb[0], rb[1], hi, lo);
    *reinterpret_cast<u16x8*>(base + 2 * kGTileElems + soff) = hi;
    *reinterpret_cast<u16x8*>(base + 3 * kGTileElems + soff) = lo;
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
      xl[kt] = tn_frag(base + 3 * kGTileElems, wk * 32 + kt * 16, fi, fg);
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const u16x8 yh = tn_frag(base, wn * 64 + nt * 16, fi, fg);
      const u16x8 yl = tn_frag(base + kGTileElems, wn * 64 + nt * 16, fi, fg);
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) acc[nt][kt] = mfma3(xh[kt], xl[kt], yh, yl, acc[nt][kt]);   // D[k = 4 fg + r][n = fi]
    }
    if (s + 1 < nsteps) lstore(cur ^ 1);
    __syncthreads();
  }
  // partial tile: lane (n = fi, 4 consecutive k) -> 16-byte stores into part[split][n][k]
  float* const out = part + (size_t)blockIdx.y * N * ld.c;     // (ld.c != K only with a single split: the einsum form below)
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const int n = n0 + wn * 64 + nt * 16 + fi, k = k0 + wk * 32 + kt * 16 + 4 * fg;
      if (n < N && k < K) *reinterpret_cast<float4*>(out + (size_t)n * ld.c + k) = float4{acc[nt][kt][0], acc[nt][kt][1], acc[nt][kt][2], acc[nt][kt][3]};
    }
  if (do_bias) {                 // (the loop's last barrier is behind every wave: the operand stages are free)
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
