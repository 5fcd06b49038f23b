// C[M,N] = epilogue(A[M,K] B[N,K]^T) on fp32 operands in split precision ("bf16x3" / three-piece): the GEMM of the training tier's
// Linear layers (axvs_train_gemm.h has the story) in a header of its own, because the inference path uses it too for projections
// whose fp32 inputs and outputs make a 128 x 128 tile pay (the deformable attention's offset / weight projection, axvs_api.hip).
// Everything here is a template or inline: the header may be included by several translation units -- but each of them must
// instantiate the kernel under its OWN tag (template parameter TU): two code objects of one library that both carry a kernel of the
// same mangled name share one host stub, and the HIP runtime aborts at the first launch through it.
#pragma once
#include <type_traits>

#include "axvs_common.h"

namespace axvs {
namespace tr {

__host__ __device__ __forceinline__ unsigned fmix32(unsigned h) {
  h ^= h >> 16;
  h *= 0x85ebca6bu;
  h ^= h >> 13;
  h *= 0xc2b2ae35u;
  h ^= h >> 16;
  return h;
}

// keep element `idx` of dropout site `site` iff (hash >> 8) >= thr, thr = floor(p * 2^24)
struct Drop {
  unsigned seed, site, thr;
  float scale;   // 1 / (1 - p)
};

__host__ __device__ __forceinline__ unsigned drop_hash(unsigned seed, unsigned site, unsigned long long idx) {
  unsigned h = seed ^ (site * 0x9E3779B9u);
  h = fmix32(h ^ (unsigned)idx);
  h = fmix32(h ^ (unsigned)(idx >> 32));
  return h;
}

__device__ __forceinline__ float drop_keep(const Drop& d, unsigned long long idx) {   // 0 or 1/(1-p)
  if (d.thr == 0) return 1.f;
  return (drop_hash(d.seed, d.site, idx) >> 8) >= d.thr ? d.scale : 0.f;
}

constexpr int kGT = 128;                 // block tile (both output dimensions)
constexpr int kGK = 32;                  // contraction step
constexpr int kGTileElems = kGT * kGK;   // one 16-bit operand tile
constexpr int kGLd = kGT + 4;            // fp32 row stride of the epilogue staging tile
constexpr size_t kGemmLds = (size_t)kGT * kGLd * sizeof(float);   // 67.6 KB >= 2 stages x 4 tiles x 8 KB

struct GemmLd {     // row strides (floats, multiples of 4) of A, B, C; ksteps > 0: split-K, that many 32-wide k-steps per blockIdx.z
  long long a, b, c;
  int ksteps;
  const float* a2 = nullptr;   // optional second A operand with A's shape and stride, added element-wise on load (x + pos)
  // alignment (in floats: 4, 2 or 1) every row start of A, B, C is known to have.  4: 16-byte vector accesses (the rule: row
  // strides and extents multiples of 4); smaller: rows of pixels whose count is whatever the image size gives (the mask einsum
  // over 193 x 337 maps) -- narrower accesses, and an extent that is not a multiple of 4 ends inside a group of four
  int al_a = 4, al_b = 4, al_c = 4;
  // tr_gemm_tn_kernel<.., STATS>: the output tile's sum (x - *stat_shift) and sum (x - *stat_shift)^2 go to
  // stat_part[(g stat_nblk + stat_blk0 + tile)][2], g = first row of the tile / stat_rows (a tile lies inside one group of rows) --
  // the one-channel BatchNorm statistics of the mask logits without a pass of their own over them
  // tr_gemm_nt_kernel<.., AFF>: the A operand is an affine function of A and a2, per group of aff_rows rows:
  // A_eff = aff[3 g] A + aff[3 g + 1] a2 + aff[3 g + 2], g = row / aff_rows (the input gradient of a one-channel BatchNorm formed in the
  // loader of the GEMM that consumes it, instead of a pass that writes it out)
  const float* aff = nullptr;
  int aff_rows = 1;
  float* stat_part = nullptr;
  const float* stat_shift = nullptr;
  int stat_nblk = 0, stat_blk0 = 0, stat_rows = 0;
};

inline bool gemm_nt_general(const GemmLd& ld, int K) { return !(ld.al_a == 4 && ld.al_b == 4 && (K & 3) == 0); }

struct GemmEpi {
  const float* bias;   // nullable [N]: added first
  float mul;           // then multiplied
  int relu;            // then max(., 0)
  Drop dr;             // then dropout, element index row * N + col (thr = 0: none)
  float beta;          // 0: overwrite C, 1: add to it
  const float* res = nullptr;   // optional residual [M][ldc], added last (out = ... + res)
  const float* res2 = nullptr;  // a second one (d_in = dv Wv + d_out + da in one epilogue)
  // out16 != nullptr: the result goes out as ONE 16-bit piece in the blocked activation layout [N/32][M][32] (kind16: 1 f16, 2 bf16)
  // instead of fp32 rows; rows flagged in zero_rows (nullable, one byte per row) are written as zeros
  u16* out16 = nullptr;
  int kind16 = 0;
  const unsigned char* zero_rows = nullptr;
};

// ---- C[M,N] = epilogue(A[M,K] . B[N,K]^T): both operands row-major with the contraction index contiguous -----------------------------
// NS = 2: the bf16x3 product above.  NS = 3: operands split into THREE bf16 pieces (24 mantissa bits = all of fp32) and six MFMAs per
// product (hh, hm, mh, mm, hl, lh; the dropped terms are 2^-24 of the product): as accurate as an fp32 GEMM at twice the matrix
// time of NS = 2.  The FORWARD GEMMs run this way (option `train_exact`, default 1).  Why: the layer has one discontinuity, the ReLU.
// With 1.5e-5 relative error on the pre-activations about one hidden unit in 1e5 lands on the other side of zero than in an fp32
// forward, and every such flip moves the input gradient of its token by ~1e-2 of the gradient's scale (measured at [1,4,256,32,32],
// d_ffn 1024, two-piece forward: d_src 7e-3 .. 2e-2 in max-norm, relative L2 3e-4, output 5e-6).  The three-piece forward costs
// 0.06 ms of a 3.3 ms step (these GEMMs are bound by their fp32 operand traffic, not by the matrix pipe), so it is the default;
// the backward GEMMs are smooth in their operands and stay two-piece.
// NS = 1 (option `train_amp`, set by the Python layer under torch.autocast): ONE 16-bit piece per operand -- the products torch's
// autocast gives the reference's nn.Linear layers (16-bit operands, fp32 accumulation); F16 selects fp16 pieces (autocast's default
// dtype on GPUs; the caller scales the loss as with the reference) instead of bf16.
template <int NS, bool F16 = false>
__device__ __forceinline__ void split_n(const float4& a, const float4& b, u16x8 (&p)[NS]) {
  const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float r = x[i];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if constexpr (F16) {
        p[s][i] = __builtin_bit_cast(u16, (_Float16)r);
      } else {
        const __bf16 h = (__bf16)r;
        p[s][i] = __builtin_bit_cast(u16, h);
        r -= (float)h;                       // exact in fp32
      }
    }
  }
}

template <int NS>
constexpr size_t gemm_nt_lds() {
  return (size_t)2 * 2 * NS * kGTileElems * sizeof(u16) > kGemmLds ? (size_t)2 * 2 * NS * kGTileElems * sizeof(u16) : kGemmLds;
}

// GEN: the general loader (rows at any 4-byte boundary, K not a multiple of 4, the x + pos addend) -- its own instantiation, so the
// kernel of the common case carries none of that code (these kernels are sensitive to their size: +3 us per launch with both
// loaders in one body)
// ADD (with !GEN): the x + pos addend ld.a2 on the 16-byte path (the general loader takes it at run time).
template <int NS, int TU = 0, bool GEN = false, bool ADD = false, bool F16 = false, bool AFF = false>
__global__ __launch_bounds__(512, NS <= 2 ? 4 : 2) void tr_gemm_nt_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                                         float* __restrict__ C, long long M, int N, int K, GemmLd ld, GemmEpi ep) {
  extern __shared__ __attribute__((aligned(16))) char gsmem[];
  u16* const sbuf = reinterpret_cast<u16*>(gsmem);            // [2 stages][A pieces | B pieces][128 rows][32] (rows chunk-swizzled)
  constexpr int kStage = 2 * NS * kGTileElems;
#ifdef AXVS_STAMPS_TR      // diagnostic build (tools/build_diag.py): phase stamps of the LAST launch, workgroups 0 .. 7
  AXVS_STAMP_DECL;
  AXVS_STAMP(0);
#endif
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
  const float* const a2p = ld.a2 ? ld.a2 + (ap - A) : nullptr;
  float af1 = 1.f, af2 = 0.f, af3 = 0.f;                      // AFF: this thread's row group
  if constexpr (AFF) {
    const float* cf = ld.aff + 3 * (int)((m0 + (a_ok ? sr : 0)) / ld.aff_rows);
    af1 = cf[0]; af2 = cf[1]; af3 = cf[2];
  }
  auto gload = [&](int ks, auto slot_tag) {
    constexpr int SL = decltype(slot_tag)::value;
    if constexpr (!GEN) {                                     // 16-byte rows, K % 4 == 0, no addend: a float4 is inside or outside
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const bool kin = kbase + ks * kGK + sq * 8 + 4 * h < K;
        ra[SL][h] = a_ok && kin ? *reinterpret_cast<const float4*>(ap + ks * kGK + 4 * h) : float4{0.f, 0.f, 0.f, 0.f};
        rb[SL][h] = b_ok && kin ? *reinterpret_cast<const float4*>(bp + ks * kGK + 4 * h) : float4{0.f, 0.f, 0.f, 0.f};
        if constexpr (ADD || AFF) {
          const float4 t = a_ok && kin ? *reinterpret_cast<const float4*>(a2p + ks * kGK + 4 * h) : float4{0.f, 0.f, 0.f, 0.f};
          if constexpr (AFF) {
            const float c3 = a_ok && kin ? af3 : 0.f;
            ra[SL][h].x = af1 * ra[SL][h].x + af2 * t.x + c3; ra[SL][h].y = af1 * ra[SL][h].y + af2 * t.y + c3;
            ra[SL][h].z = af1 * ra[SL][h].z + af2 * t.z + c3; ra[SL][h].w = af1 * ra[SL][h].w + af2 * t.w + c3;
          } else {
            ra[SL][h].x += t.x; ra[SL][h].y += t.y; ra[SL][h].z += t.z; ra[SL][h].w += t.w;
          }
        }
      }
    } else {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int kv = K - (kbase + ks * kGK + sq * 8 + 4 * h);   // how many of the four columns exist
        ra[SL][h] = ldg4(ap + ks * kGK + 4 * h, a_ok ? kv : 0, ld.al_a);
        rb[SL][h] = ldg4(bp + ks * kGK + 4 * h, b_ok ? kv : 0, ld.al_b);
        if constexpr (AFF) {
          const int nv = a_ok ? kv : 0;
          const float4 t = ldg4(a2p + ks * kGK + 4 * h, nv, ld.al_a);
          ra[SL][h].x = nv > 0 ? af1 * ra[SL][h].x + af2 * t.x + af3 : 0.f; ra[SL][h].y = nv > 1 ? af1 * ra[SL][h].y + af2 * t.y + af3 : 0.f;
          ra[SL][h].z = nv > 2 ? af1 * ra[SL][h].z + af2 * t.z + af3 : 0.f; ra[SL][h].w = nv > 3 ? af1 * ra[SL][h].w + af2 * t.w + af3 : 0.f;
        } else if (ld.a2 && a_ok && kv > 0) {
          const float4 t = ldg4(ld.a2 + (ap - A) + ks * kGK + 4 * h, kv, ld.al_a);
          ra[SL][h].x += t.x; ra[SL][h].y += t.y; ra[SL][h].z += t.z; ra[SL][h].w += t.w;
        }
      }
    }
  };
  auto lstore = [&](int stage, auto slot_tag) {
    constexpr int SL = decltype(slot_tag)::value;
    u16* const base = sbuf + stage * kStage;
    u16x8 p[NS];
    split_n<NS, F16>(ra[SL][0], ra[SL][1], p);
#pragma unroll
    for (int s = 0; s < NS; ++s) *reinterpret_cast<u16x8*>(base + s * kGTileElems + soff) = p[s];
    split_n<NS, F16>(rb[SL][0], rb[SL][1], p);
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
#ifdef AXVS_STAMPS_TR
    if (ks == 2) AXVS_STAMP(8);
#endif
    if (ks + 2 < nk) gload(ks + 2, std::integral_constant<int, PAR>{});
    const u16* const base = sbuf + PAR * kStage;
    u16x8 bf[2][NS];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int s = 0; s < NS; ++s) bf[nt][s] = *reinterpret_cast<const u16x8*>(base + (NS + s) * kGTileElems + boff[nt]);
    // Piece products in the order hh, hm, mh, mm, hl, lh PER ACCUMULATOR (that order is what the result's bits depend on), but
    // issued round-robin over the four accumulators of two m-tiles: an MFMA never waits for the one before it.
    constexpr int kPieceA[6] = {0, 1, 0, 1, 2, 0}, kPieceB[6] = {0, 0, 1, 1, 0, 2};   // (af piece, bf piece) of product p
    constexpr int kProducts = NS == 1 ? 1 : NS == 2 ? 3 : 6;
#pragma unroll
    for (int mp = 0; mp < 2; ++mp) {
      u16x8 af[2][NS];
#pragma unroll
      for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
        for (int s = 0; s < NS; ++s) af[m2][s] = *reinterpret_cast<const u16x8*>(base + s * kGTileElems + aoff[2 * mp + m2]);
#pragma unroll
      for (int p = 0; p < kProducts; ++p)
#pragma unroll
        for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {                      // D[n = 4 fg + r][m = fi]
            if (p == 0) acc[2 * mp + m2][nt] = H16<!F16>::mfma(bf[nt][0], af[m2][0], acc[2 * mp + m2][nt]);
            else acc[2 * mp + m2][nt] = H16<true>::mfma(bf[nt][kPieceB[p]], af[m2][kPieceA[p]], acc[2 * mp + m2][nt]);
          }
    }
    // the next step's operands into the other stage (every wave left its reads of it behind the previous barrier) -- unconditionally:
    // after the last step the registers hold an old tile and the stage is not read again.  (Forcing the split's VALU instructions
    // between the MFMAs with sched_group_barrier -- 1 MFMA : 2 VALU -- was measured: k-loop 24.0 k -> 27.6 k cycles, not kept.)
#ifdef AXVS_STAMPS_TR
    if (ks == 2) AXVS_STAMP(9);                  // fragments read, MFMAs issued
#endif
    lstore(PAR ^ 1, std::integral_constant<int, PAR ^ 1>{});
#ifdef AXVS_STAMPS_TR
    if (ks == 2) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      AXVS_STAMP(10);                            // next tile split and stored (its global loads waited for)
    }
#endif
    __syncthreads();
#ifdef AXVS_STAMPS_TR
    if (ks == 2) AXVS_STAMP(11);                 // barrier passed
#endif
  };
  if (nk > 0) gload(0, S0{});
  if (nk > 1) gload(1, S1{});
  if (nk > 0) lstore(0, S0{});
  __syncthreads();
#ifdef AXVS_STAMPS_TR
  AXVS_STAMP(1);
#endif
  for (int ks = 0; ks < nk; ks += 2) {
    kstep(ks, S0{});
    if (ks + 1 < nk) kstep(ks + 1, S1{});
  }
#ifdef AXVS_STAMPS_TR
  AXVS_STAMP(2);
#endif
  // ---- epilogue: accumulators -> fp32 staging tile [m][n] -> whole rows, 512 bytes per row segment ----
  float* const stg = reinterpret_cast<float*>(gsmem);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
      *reinterpret_cast<float4*>(stg + (wm * 64 + mt * 16 + fi) * kGLd + wn * 32 + nt * 16 + 4 * fg) =
          float4{acc[mt][nt][0], acc[mt][nt][1], acc[mt][nt][2], acc[mt][nt][3]};
  __syncthreads();
#ifdef AXVS_STAMPS_TR
  AXVS_STAMP(3);
#endif
  // The common epilogues (nothing, + bias, * mul) without the per-round branches of the general form below: the eight LDS reads of
  // a thread first, then eight stores (the stamps put the general form at 675 cycles per round: ten uniform branches and their
  // waits between an LDS read and its store; this one at a third of that).
  const bool plain_epi = !ep.relu && ep.dr.thr == 0 && !ep.out16 && ep.beta == 0.f && !ep.res && !ep.res2;
  if (plain_epi) {
    const int c4 = tid & 31, gn = n0 + 4 * c4;
    float4 b = {0.f, 0.f, 0.f, 0.f};
    if (ep.bias && gn < N) b = *reinterpret_cast<const float4*>(ep.bias + gn);
    float4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const float4*>(stg + ((tid >> 5) + 16 * i) * kGLd + 4 * c4);
    const float mul = ep.mul;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const long long gm = m0 + (tid >> 5) + 16 * i;
      if (gm < M && gn < N)
        *reinterpret_cast<float4*>(C + gm * ld.c + gn) = make_float4((v[i].x + b.x) * mul, (v[i].y + b.y) * mul, (v[i].z + b.z) * mul, (v[i].w + b.w) * mul);
    }
  } else if (!ep.out16) {
    // fp32 rows with ReLU / dropout / accumulate / residuals: the same arithmetic as the loop below, but per half (four rounds) every
    // load -- LDS, C for beta, the residuals -- is issued before the first use and each uniform condition is tested once per half
    const int c4 = tid & 31, gn = n0 + 4 * c4;
    const bool cin = gn < N, drop = ep.dr.thr != 0, relu = ep.relu != 0;
    float4 b = {0.f, 0.f, 0.f, 0.f};
    if (ep.bias && cin) b = *reinterpret_cast<const float4*>(ep.bias + gn);
    const float mul = ep.mul;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      float4 v[4], o[4], r1[4], r2[4];
      bool ok[4];
      long long off[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = (tid >> 5) + 16 * (4 * hf + i);
        ok[i] = cin && m0 + row < M;
        off[i] = (m0 + row) * ld.c + gn;
        v[i] = *reinterpret_cast<const float4*>(stg + row * kGLd + 4 * c4);
        o[i] = r1[i] = r2[i] = float4{0.f, 0.f, 0.f, 0.f};
      }
      if (ep.beta != 0.f) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (ok[i]) o[i] = *reinterpret_cast<const float4*>(C + off[i]);
      }
      if (ep.res) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (ok[i]) r1[i] = *reinterpret_cast<const float4*>(ep.res + off[i]);
      }
      if (ep.res2) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (ok[i]) r2[i] = *reinterpret_cast<const float4*>(ep.res2 + off[i]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float t[4] = {(v[i].x + b.x) * mul, (v[i].y + b.y) * mul, (v[i].z + b.z) * mul, (v[i].w + b.w) * mul};
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] = relu ? fmaxf(t[e], 0.f) : t[e];
        if (drop) {
          const unsigned long long e0 = (unsigned long long)(m0 + (tid >> 5) + 16 * (4 * hf + i)) * N + gn;
#pragma unroll
          for (int e = 0; e < 4; ++e) t[e] *= drop_keep(ep.dr, e0 + e);
        }
        // (the sums in the order of the loop below: + C, + res, + res2)
        const float ox[4] = {o[i].x, o[i].y, o[i].z, o[i].w}, ax[4] = {r1[i].x, r1[i].y, r1[i].z, r1[i].w}, bx[4] = {r2[i].x, r2[i].y, r2[i].z, r2[i].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (ep.beta != 0.f) t[e] += ox[e];
          if (ep.res) t[e] += ax[e];
          if (ep.res2) t[e] += bx[e];
        }
        if (ok[i]) *reinterpret_cast<float4*>(C + off[i]) = make_float4(t[0], t[1], t[2], t[3]);
      }
    }
  } else
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
      if (ep.out16) {
        if (ep.zero_rows && ep.zero_rows[gm]) t[0] = t[1] = t[2] = t[3] = 0.f;
        typedef unsigned short u16x4v __attribute__((ext_vector_type(4)));
        u16x4v o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = ep.kind16 == 2 ? H16<true>::from_f32(t[e]) : H16<false>::from_f32(t[e]);
        *reinterpret_cast<u16x4v*>(ep.out16 + blk_off(M, gm, gn)) = o;
        continue;
      }
      if (ep.beta != 0.f) {
        const float4 o = *reinterpret_cast<const float4*>(cp);
        t[0] += o.x; t[1] += o.y; t[2] += o.z; t[3] += o.w;
      }
      if (ep.res) {
        const float4 o = *reinterpret_cast<const float4*>(ep.res + gm * ld.c + gn);
        t[0] += o.x; t[1] += o.y; t[2] += o.z; t[3] += o.w;
      }
      if (ep.res2) {
        const float4 o = *reinterpret_cast<const float4*>(ep.res2 + gm * ld.c + gn);
        t[0] += o.x; t[1] += o.y; t[2] += o.z; t[3] += o.w;
      }
      *reinterpret_cast<float4*>(cp) = make_float4(t[0], t[1], t[2], t[3]);
    }
  }
#ifdef AXVS_STAMPS_TR
  AXVS_STAMP(4);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  AXVS_STAMP(5);
  if constexpr (TU == 0) {
    AXVS_STAMP_FLUSH(6);
    if (blockIdx.x < 8 && (threadIdx.x & 63) == 0)
      for (int i_ = 8; i_ < 12; ++i_) ::axvs::g_stamps[(i_ + 2) * 64 + (blockIdx.x & 7) * 8 + (threadIdx.x >> 6)] = st_[i_];
    if (blockIdx.x == 0 && threadIdx.x == 0) {           // slot 6: the launch's shape
      ::axvs::g_stamps[6 * 64 + 0] = (unsigned long long)M;
      ::axvs::g_stamps[6 * 64 + 1] = (unsigned long long)N;
      ::axvs::g_stamps[6 * 64 + 2] = (unsigned long long)K;
      ::axvs::g_stamps[6 * 64 + 3] = (unsigned long long)NS;
      ::axvs::g_stamps[6 * 64 + 4] = (unsigned long long)(ep.res != nullptr) + 2 * (ep.bias != nullptr) + 4 * (ep.dr.thr != 0);
    }
  }
#endif
}

}  // namespace tr
}  // namespace axvs
