// Token-GEMM  Y[m, n] = sum_k X[m, k] * W[n, k]  (+bias, epilogue), 16-bit MFMA operands, fp32 accumulate.
//
// X comes from a loader functor (fp32 token rows gathered through a RowMap, or a blocked 16-bit
// activation matrix), W is a pre-packed blocked 16-bit weight [K/32][Nout][32].  The MFMA is issued
// "transposed" -- weights as the A operand, activations as the B operand -- so that every lane ends up
// with 4 consecutive output channels of one token (D[n][m]: m = lane&15, n = 4*(lane>>4)+r): 8-byte
// (16-bit) or 16-byte (fp32) stores per lane.
#pragma once
#include <type_traits>
#include "axvs_common.h"

namespace axvs {

// ---------------- loaders: 8 consecutive k of token row m, converted to the 16-bit operand type ----------------
struct F32x8 { float4 a, b; };

// what a prefetching consumer keeps in flight for a loader: its raw_t (fetch now, conv at use) if it has one, else the operand
template <class A, class = void>
struct RawOf {
  typedef u16x8 type;
  static __device__ __forceinline__ type fetch(const A& a, int m, int k) { return a.load(m, k); }
  static __device__ __forceinline__ u16x8 conv(const type& r) { return r; }
};
template <class A>
struct RawOf<A, std::void_t<typename A::raw_t>> {
  typedef typename A::raw_t type;
  static __device__ __forceinline__ type fetch(const A& a, int m, int k) { return a.fetch(m, k); }
  static __device__ __forceinline__ u16x8 conv(const type& r) { return A::conv(r); }
};

template <bool BF>
struct ALoadRowsF32 {
  static constexpr int kPrefetch = 3;
  const float* src;   // [rows, K] fp32
  const float* add;   // nullable: added element-wise (positional embedding)
  RowMap rm;
  int M, K;
  __device__ __forceinline__ u16x8 load(int m, int k) const { return conv(fetch(m, k)); }
  struct raw_t { float4 a, b, c, d; };      // row words and (optional) positional words, summed and rounded at the point of use
  __device__ __forceinline__ raw_t fetch(int m, int k) const {
    m = min(m, M - 1);
    long long off = nat_row(rm, m) * K + k;
    const float4* p = reinterpret_cast<const float4*>(src + off);
    raw_t r{p[0], p[1], float4{0.f, 0.f, 0.f, 0.f}, float4{0.f, 0.f, 0.f, 0.f}};
    if (add) {
      const float4* q = reinterpret_cast<const float4*>(add + off);
      r.c = q[0];
      r.d = q[1];
    }
    return r;
  }
  static __device__ __forceinline__ u16x8 conv(const raw_t& r) {
    float v[8] = {r.a.x + r.c.x, r.a.y + r.c.y, r.a.z + r.c.z, r.a.w + r.c.w, r.b.x + r.d.x, r.b.y + r.d.y, r.b.z + r.d.z, r.b.w + r.d.w};
    return cvt8<BF>(v);
  }
};

// fp32 rows with an explicit leading dimension / column offset (a column slice of a wider matrix), identity row order
template <bool BF>
struct ALoadRowsLd {
  static constexpr int kPrefetch = 3;
  const float* src;
  int ld, col0, M;
  __device__ __forceinline__ u16x8 load(int m, int k) const { return conv(fetch(m, k)); }
  // fetch / conv split: a prefetching consumer keeps the raw fp32 words in flight and converts at the point of use (converting
  // at the load would make every prefetch wait for its own data)
  typedef F32x8 raw_t;
  __device__ __forceinline__ raw_t fetch(int m, int k) const {
    m = min(m, M - 1);
    const float4* p = reinterpret_cast<const float4*>(src + (long long)m * ld + col0 + k);
    return raw_t{p[0], p[1]};
  }
  static __device__ __forceinline__ u16x8 conv(const raw_t& r) {
    float v[8] = {r.a.x, r.a.y, r.a.z, r.a.w, r.b.x, r.b.y, r.b.z, r.b.w};
    return cvt8<BF>(v);
  }
};

// The folded temporal ASPP as a GEMM (axvs_cc.h, pack_aspp_taps_kernel): rows are (bq, t) of a [BQ, Tc, 256] fp32 tensor;
// k = j * 256 + ci reads channel ci of row (bq, clamp(t + off[j])) -- replicate padding (CC/maxtron_cross_clip_tracking_module.py:180-182).
template <bool BF>
struct ALoadTaps7 {
  static constexpr int kPrefetch = 3;
  const float* src;
  int Tc, M;
  int off[7];
  __device__ __forceinline__ u16x8 load(int m, int k) const { return conv(fetch(m, k)); }
  typedef F32x8 raw_t;
  __device__ __forceinline__ raw_t fetch(int m, int k) const {
    m = min(m, M - 1);
    const int j = k >> 8, ci = k & 255;
    const int bq = m / Tc, t = m - bq * Tc;
    const int tt = min(max(t + off[j], 0), Tc - 1);
    const float4* p = reinterpret_cast<const float4*>(src + ((long long)bq * Tc + tt) * 256 + ci);
    return raw_t{p[0], p[1]};
  }
  static __device__ __forceinline__ u16x8 conv(const raw_t& r) {
    float v[8] = {r.a.x, r.a.y, r.a.z, r.a.w, r.b.x, r.b.y, r.b.z, r.b.w};
    return cvt8<BF>(v);
  }
};

template <bool BF>
struct ALoadBlocked {
  static constexpr int kPrefetch = 3;
  const u16* X;       // blocked [K/32][R][32]
  long long R;
  int M;
  int diagT, diagN, diagL;  // diagT > 0: row m -> frame(m)*M + m   (own-frame slot of the [T][M] T-expanded tensor)
  __device__ __forceinline__ u16x8 load(int m, int k) const {
    m = min(m, M - 1);
    long long r = m;
    if (diagT > 0) r = (long long)((m % diagN) / diagL) * M + m;
    return *reinterpret_cast<const u16x8*>(X + blk_off(R, r, k));
  }
};

// ---- split-precision operands: x = hi + lo (both 16-bit), W = hi + lo  =>  x.W ~ hi.hi + hi.lo + lo.hi as ONE GEMM over a
//      3x longer K:  activations (hi | hi | lo)  against weights packed as (hi | lo | hi) (pack_weight_split3_kernel).  Used
//      where a raw projection output (no residual / norm behind it) has to hold the 1e-3 bar: ~16 mantissa bits for 3x the
//      MFMA work of a small GEMM. ----
template <bool BF>
struct ALoadRowsF32Split3 {
  static constexpr int kPrefetch = 1;
  const float* src;   // [M, K] fp32 rows
  int M, K;
  const float* add = nullptr;   // optional [M, K], added element-wise (positional embedding)
  __device__ __forceinline__ u16x8 load(int m, int k) const {
    m = min(m, M - 1);
    const int part = k / K, kk = k - part * K;
    const float4* p = reinterpret_cast<const float4*>(src + (long long)m * K + kk);
    float4 a = p[0], b = p[1];
    if (add) {
      const float4* q = reinterpret_cast<const float4*>(add + (long long)m * K + kk);
      const float4 c = q[0], d = q[1];
      a.x += c.x; a.y += c.y; a.z += c.z; a.w += c.w;
      b.x += d.x; b.y += d.y; b.z += d.z; b.w += d.w;
    }
    float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    u16x8 hi = cvt8<BF>(v);
    if (part < 2) return hi;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] -= H16<BF>::to_f32(hi[i]);
    return cvt8<BF>(v);
  }
};

template <bool BF>
struct ALoadBlockedSplit3 {
  static constexpr int kPrefetch = 1;
  const u16* X;       // blocked [2*Kp/32][R][32]: hi blocks, then lo blocks
  long long R;
  int M, Kp;
  __device__ __forceinline__ u16x8 load(int m, int k) const {
    m = min(m, M - 1);
    const int part = k / Kp, kk = k - part * Kp;
    return *reinterpret_cast<const u16x8*>(X + blk_off(R, m, (part == 2 ? Kp : 0) + kk));
  }
};

// ---------------- epilogues: 4 consecutive output channels n..n+3 of token m ----------------

template <bool BF>
struct EpiBlocked16 {
  u16* Y;             // blocked [Nout/32][R][32]
  long long R;
  const float* bias;  // [Nout]
  float scale;        // applied to channels < nscale after the bias (softmax scale folded into q)
  int nscale;
  int relu;           // activation: 0 none, 1 ReLU, 2 exact GELU
  const float* mul = nullptr;   // optional per-channel multiplier applied to the accumulator before the bias (folded BN)
  int n_off = 0;                // column offset in Y (concatenating several GEMMs along channels)
  const unsigned char* zero_rows = nullptr;   // optional: rows with a non-zero flag are written as zeros (padding masks)
  __device__ __forceinline__ void store(int m, int n, f32x4 v) const {
    if (mul) {
      float4 s = *reinterpret_cast<const float4*>(mul + n);
      v[0] *= s.x; v[1] *= s.y; v[2] *= s.z; v[3] *= s.w;
    }
    float4 b = *reinterpret_cast<const float4*>(bias + n);
    v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
    n += n_off;
    if (n < nscale) v *= scale;
    if (relu == 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
    } else if (relu == 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = gelu_exact(v[i]);
    }
    if (zero_rows && zero_rows[m]) v = f32x4{0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<u16x4*>(Y + blk_off(R, m, n)) = cvt4<BF>(v);
  }
};

struct EpiRowsF32 {
  float* Y;           // [rows, ld] fp32, rows through rm
  const float* res;   // nullable residual, same indexing as Y
  const float* bias;  // nullable
  RowMap rm;
  int ld;
  float scale;
  const float* mul = nullptr;   // optional per-channel multiplier (folded BN)
  int gelu = 0;
  __device__ __forceinline__ void store(int m, int n, f32x4 v) const {
    if (mul) {
      float4 s = *reinterpret_cast<const float4*>(mul + n);
      v[0] *= s.x; v[1] *= s.y; v[2] *= s.z; v[3] *= s.w;
    }
    float4 b = bias ? *reinterpret_cast<const float4*>(bias + n) : float4{0.f, 0.f, 0.f, 0.f};
    long long off = nat_row(rm, m) * ld + n;
    float4 o = {(v[0] + b.x) * scale, (v[1] + b.y) * scale, (v[2] + b.z) * scale, (v[3] + b.w) * scale};
    if (gelu) o = float4{gelu_exact(o.x), gelu_exact(o.y), gelu_exact(o.z), gelu_exact(o.w)};
    if (res) {
      float4 r = *reinterpret_cast<const float4*>(res + off);
      o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
    }
    *reinterpret_cast<float4*>(Y + off) = o;
  }
};

// ---------------- v1 kernel: 64x64 tile, 4 waves (2x2), both operands staged through LDS ----------------
template <bool BF, class ALoad, class Epi>
__device__ __forceinline__ void gemm64_body(const ALoad& al, const u16* __restrict__ Wp, const Epi& epi, int M, int Nout, int K) {
  __shared__ __attribute__((aligned(16))) u16 sX[2][64 * 32];      // two buffers, used alternately: one barrier per k-step
  __shared__ __attribute__((aligned(16))) u16 sW[2][64 * 32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
  const int lrow = tid >> 2, lg = tid & 3;                 // staging: one 16-byte chunk per thread and operand
  const int st_off = lrow * 32 + swz_chunk(lrow, lg) * 8;
  // weights: thread t of a wave takes bytes [16 t, 16 t + 16) of its (k-block, 16 rows) block (wblk_off: fragment order)
  const int wlrow = wave * 16 + (lane & 15), wlg = lane >> 4;
  const int wrow = min(n0 + wlrow, Nout - 1);
  const int wst_off = wlrow * 32 + swz_chunk(wlrow, wlg) * 8;
  const int fi = lane & 15, fg = lane >> 4;

  f32x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nkb = K >> 5;
  // global -> register prefetch depth in k-steps: 3 hides the L2 latency of the small, latency-bound GEMMs (cross-clip
  // module: -12 %); loaders that do split-precision arithmetic on fp32 rows are throughput-bound on big grids and lose
  // occupancy to the extra registers, they declare kPrefetch = 1
  constexpr int PF = ALoad::kPrefetch;
  typedef RawOf<ALoad> Raw;                 // fp32-source loaders keep the raw words in flight and convert when staging
  typename Raw::type rx[PF];
  u16x8 rw[PF];
#pragma unroll
  for (int u = 0; u < PF; ++u) {
    const int k = min(u, nkb - 1) * 32 + lg * 8;
    rx[u] = Raw::fetch(al, m0 + lrow, k);
    rw[u] = *reinterpret_cast<const u16x8*>(Wp + wblk_off(Nout, wrow, k - lg * 8 + wlg * 8));
  }
  for (int kb0 = 0; kb0 < nkb; kb0 += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int kb = kb0 + u;
      if (kb < nkb) {                       // uniform
        u16* bx = sX[kb & 1];
        u16* bw = sW[kb & 1];
        *reinterpret_cast<u16x8*>(bx + st_off) = Raw::conv(rx[u]);
        *reinterpret_cast<u16x8*>(bw + wst_off) = rw[u];
        __syncthreads();                    // also orders these writes after the reads of this buffer two steps ago
        {
          const int k = min(kb + PF, nkb - 1) * 32 + lg * 8;   // unconditional (clamped) prefetch keeps vmcnt bookkeeping simple
          rx[u] = Raw::fetch(al, m0 + lrow, k);
          rw[u] = *reinterpret_cast<const u16x8*>(Wp + wblk_off(Nout, wrow, k - lg * 8 + wlg * 8));
        }
        u16x8 fx[2], fw[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          int r = wm * 32 + i * 16 + fi;
          fx[i] = *reinterpret_cast<const u16x8*>(bx + r * 32 + swz_chunk(r, fg) * 8);
          int c = wn * 32 + i * 16 + fi;
          fw[i] = *reinterpret_cast<const u16x8*>(bw + c * 32 + swz_chunk(c, fg) * 8);
        }
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) acc[ni][mi] = H16<BF>::mfma(fw[ni], fx[mi], acc[ni][mi]);
      }
    }
  }
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      int m = m0 + wm * 32 + mi * 16 + fi;
      int n = n0 + wn * 32 + ni * 16 + fg * 4;
      if (m < M && n < Nout) epi.store(m, n, acc[ni][mi]);
    }
}

template <bool BF, class ALoad, class Epi>
__global__ __launch_bounds__(256) void gemm64_kernel(ALoad al, const u16* __restrict__ Wp, Epi epi, int M, int Nout, int K) {
  gemm64_body<BF>(al, Wp, epi, M, Nout, K);
}

// ---------------- small problems: one wave per 16 x (16 NT) tile, operands straight from L2 to VGPRs ----------------
// A GEMM with a few hundred rows (the cross-clip module: 512 clip queries) gives the 64x64 kernel only a handful of workgroups,
// each walking K in dependent global -> LDS -> barrier -> MFMA steps (~0.5 us per k-step measured).  Here every wave owns a
// 16-row x 16*NT-column tile and streams its operand fragments with D k-blocks in flight -- both blocked layouts make a fragment
// 1 KiB contiguous --, no LDS, no barriers; the redundant activation loads of the waves sharing a row tile hit L1 / L2.
// NFAST: the column tiles are the fast grid dimension -- workgroups b and b + 8 share an XCD (round-robin placement; speed only), so with
// 8 column tiles every XCD's L2 fetches ONE column slice of the weights instead of all of them (cold weights: the cross-clip chain)
template <bool BF, class ALoad, class Epi, int NT, bool NFAST = false>
__device__ __forceinline__ void gemm_direct_body(const ALoad& al, const u16* __restrict__ Wp, const Epi& epi, int M, int Nout, int K) {
  constexpr int D = 8;
  const int lane = threadIdx.x & 63, fi = lane & 15, fg = lane >> 4;
  const int m0 = (NFAST ? blockIdx.y : blockIdx.x) * 16, n0 = (NFAST ? blockIdx.x : blockIdx.y) * 16 * NT;
  // split-K over the waves of the workgroup (1..4; the launcher picks K / 256 when that divides): the loop below is bound by
  // L2 round trips -- one per group of D k-blocks --, so K = 768 as three waves with 8 k-blocks each costs one round trip
  // instead of three; the partial tiles meet in LDS
  const int ks = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const int nkb = (K >> 5) / ks, kbase = wave * nkb;
  int nrow[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) nrow[nt] = min(n0 + nt * 16 + fi, Nout - 1);
  typedef RawOf<ALoad> Raw;
  typename Raw::type xb[D];
  u16x8 wa[D][NT];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const int k = (kbase + min(d, nkb - 1)) * 32 + fg * 8;
    xb[d] = Raw::fetch(al, m0 + fi, k);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) wa[d][nt] = *reinterpret_cast<const u16x8*>(Wp + wblk_off(Nout, nrow[nt], k));
  }
  f32x4 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (nkb % D == 0) {
    // whole groups of D k-blocks (K a multiple of 256: every caller today): no control flow inside the group, so the loads of a
    // group are all in flight together (with the conditional form below the compiler waits for each step's loads in turn and
    // every k-step pays an L2 round trip: 13.2 -> 9.8 us for the ASPP branches, K = 768)
    for (int kb0 = 0; kb0 < nkb; kb0 += D) {
#pragma unroll
      for (int u = 0; u < D; ++u) {
        const u16x8 xo = Raw::conv(xb[u]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = H16<BF>::mfma(wa[u][nt], xo, acc[nt]);
        if (kb0 + D < nkb) {                // uniform; false throughout when the wave's share is a single group
          const int k = (kbase + min(kb0 + u + D, nkb - 1)) * 32 + fg * 8;
          xb[u] = Raw::fetch(al, m0 + fi, k);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) wa[u][nt] = *reinterpret_cast<const u16x8*>(Wp + wblk_off(Nout, nrow[nt], k));
        }
      }
    }
  } else
  for (int kb0 = 0; kb0 < nkb; kb0 += D) {
#pragma unroll
    for (int u = 0; u < D; ++u) {
      if (kb0 + u < nkb) {                  // uniform
        const u16x8 xo = Raw::conv(xb[u]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = H16<BF>::mfma(wa[u][nt], xo, acc[nt]);
        const int k = (kbase + min(kb0 + u + D, nkb - 1)) * 32 + fg * 8;   // unconditional (clamped) refill of the slot just used
        xb[u] = Raw::fetch(al, m0 + fi, k);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wa[u][nt] = *reinterpret_cast<const u16x8*>(Wp + wblk_off(Nout, nrow[nt], k));
      }
    }
  }
  if (ks > 1) {
    __shared__ f32x4 red[6][NT][64];      // up to 7 k-waves (the folded ASPP: K = 7 * 256)
    if (wave > 0) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) red[wave - 1][nt][lane] = acc[nt];
    }
    __syncthreads();
    if (wave > 0) return;
    for (int j = 0; j < ks - 1; ++j)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] += red[j][nt][lane];
  }
  const int m = m0 + fi;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = n0 + nt * 16 + fg * 4;
    if (m < M && n < Nout) epi.store(m, n, acc[nt]);
  }
}

template <bool BF, class ALoad, class Epi, int NT>
__global__ __launch_bounds__(256) void gemm_direct_kernel(ALoad al, const u16* __restrict__ Wp, Epi epi, int M, int Nout, int K) {
  gemm_direct_body<BF, ALoad, Epi, NT>(al, Wp, epi, M, Nout, K);
}
template <bool BF, class ALoad, class Epi, int NT>      // 5 .. 7 k-waves
__global__ __launch_bounds__(448) void gemm_direct_wide_kernel(ALoad al, const u16* __restrict__ Wp, Epi epi, int M, int Nout, int K) {
  gemm_direct_body<BF, ALoad, Epi, NT, true>(al, Wp, epi, M, Nout, K);
}

// 64x64 tiles unless they would leave most of the chip idle
// waves per workgroup of the direct kernels: K split into 256-wide shares when that divides (at most 4)
inline int gemm_direct_waves(int K, int max_ks = 4) {
  const int nkb = K >> 5;
  for (int ks = max_ks; ks > 1; --ks)
    if (nkb % (8 * ks) == 0) return ks;
  return 1;
}
inline int& gemm_small_upto() { static thread_local int v = 128; return v; }      // option "gemm_small_upto" (tiles of 64 x 64 outputs up to which the direct kernels run)
// (round 5: with a long reduction -- K >= 1024: the split-precision 1x1 projections of the pixel decoder -- the direct kernels, which split K over their waves, pay up
//  to twice as many tiles: the VIPSeg module's res5 projection [2150 x 256 x 6144] is 136 tiles; short reductions such as the cross-clip embeddings [2048 x 512 x 256] do not)
inline bool gemm_is_small(int M, int Nout, int nb = 1, int K = 0) {
  const long long tiles = (long long)((M + 63) / 64) * ((Nout + 63) / 64) * nb;
  return tiles <= gemm_small_upto() || (K >= 1024 && tiles <= 2 * gemm_small_upto());
}

// NB independent GEMMs of the same shape in ONE launch (blockIdx.z picks the problem): small dependent-free GEMMs such as the
// three dilated branches of the temporal ASPP fill the chip together instead of queueing behind each other.
template <class ALoad, class Epi, int NB>
struct GemmBatch {
  ALoad al[NB];
  const u16* W[NB];
  Epi epi[NB];
};
template <bool BF, class ALoad, class Epi, int NB>
__global__ __launch_bounds__(256) void gemm64_batched_kernel(GemmBatch<ALoad, Epi, NB> b, int M, int Nout, int K) {
  const int z = blockIdx.z;
  gemm64_body<BF>(b.al[z], b.W[z], b.epi[z], M, Nout, K);
}
template <bool BF, class ALoad, class Epi, int NB>
__global__ __launch_bounds__(256) void gemm_direct_batched_kernel(GemmBatch<ALoad, Epi, NB> b, int M, int Nout, int K) {
  const int z = blockIdx.z;
  gemm_direct_body<BF, ALoad, Epi, 2>(b.al[z], b.W[z], b.epi[z], M, Nout, K);
}
template <bool BF, class ALoad, class Epi, int NB>
inline void launch_gemm_batched(const GemmBatch<ALoad, Epi, NB>& b, int M, int Nout, int K, hipStream_t st) {
  if (gemm_is_small(M, Nout, NB, K)) {
    dim3 grid((M + 15) / 16, (Nout + 31) / 32, NB);
    hipLaunchKernelGGL((gemm_direct_batched_kernel<BF, ALoad, Epi, NB>), grid, dim3(64 * gemm_direct_waves(K)), 0, st, b, M, Nout, K);
    return;
  }
  dim3 grid((M + 63) / 64, (Nout + 63) / 64, NB);
  hipLaunchKernelGGL((gemm64_batched_kernel<BF, ALoad, Epi, NB>), grid, dim3(256), 0, st, b, M, Nout, K);
}

template <bool BF, class ALoad, class Epi>
inline void launch_gemm(const ALoad& al, const u16* Wp, const Epi& epi, int M, int Nout, int K, hipStream_t st, int max_ks = 4 /* k-waves of the direct kernels: up to 7 */) {
  if (gemm_is_small(M, Nout, 1, K)) {
    const int kw = gemm_direct_waves(K, max_ks);
    const bool wide_n = (long long)((M + 15) / 16) * ((Nout + 63) / 64) >= 256;
    if (kw > 4) {
      if (wide_n) hipLaunchKernelGGL((gemm_direct_wide_kernel<BF, ALoad, Epi, 4>), dim3((Nout + 63) / 64, (M + 15) / 16), dim3(64 * kw), 0, st, al, Wp, epi, M, Nout, K);
      else hipLaunchKernelGGL((gemm_direct_wide_kernel<BF, ALoad, Epi, 2>), dim3((Nout + 31) / 32, (M + 15) / 16), dim3(64 * kw), 0, st, al, Wp, epi, M, Nout, K);
    } else if (wide_n) {
      hipLaunchKernelGGL((gemm_direct_kernel<BF, ALoad, Epi, 4>), dim3((M + 15) / 16, (Nout + 63) / 64), dim3(64 * kw), 0, st, al, Wp, epi, M, Nout, K);
    } else {   // narrower tiles: twice the waves
      hipLaunchKernelGGL((gemm_direct_kernel<BF, ALoad, Epi, 2>), dim3((M + 15) / 16, (Nout + 31) / 32), dim3(64 * kw), 0, st, al, Wp, epi, M, Nout, K);
    }
    return;
  }
  dim3 grid((M + 63) / 64, (Nout + 63) / 64);
  hipLaunchKernelGGL((gemm64_kernel<BF, ALoad, Epi>), grid, dim3(256), 0, st, al, Wp, epi, M, Nout, K);
}

}  // namespace axvs
