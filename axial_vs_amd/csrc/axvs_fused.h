// Fused row-block kernels for the C = 256, heads = 8 configuration (every shipped config):
//   ffn_fused_kernel      norm1 -> linear1 -> ReLU -> linear2 -> +residual -> norm2   (WC/temporal_attention.py:181-185,217-218)
//
// Common structure ("N-split"): a workgroup owns 64 token rows; its 8 waves split the OUTPUT channels of every
// GEMM.  Activations live in LDS as [K/32][64 rows][32] 16-bit tiles (64-byte rows, chunk-swizzled so the
// ds_read_b128 fragment reads are conflict free) and are shared by all waves; every weight fragment is needed by
// exactly one wave, so weights go L2 -> VGPR directly (1 KiB coalesced loads from the blocked layout), a whole GEMM
// phase ahead of their use, and never touch LDS.  MFMA orientation is D[channel][token] (weights = A operand).
#pragma once
#include <type_traits>

#include "axvs_common.h"

namespace axvs {

constexpr int kRows = 64;                       // token rows per workgroup
constexpr int kTileElems = kRows * 32;          // one k-block of an activation tile

// element offset of 16-byte chunk `c` (0..3) of row `row` in k-block `kb`
__device__ __forceinline__ int act_off(int kb, int row, int c) { return (kb * kRows + row) * 32 + swz_chunk(row, c) * 8; }

// B-operand fragment (activations) of m-tile mt, k-block kb
__device__ __forceinline__ u16x8 act_frag(const u16* tile, int kb, int mt, int fi, int fg) {
  return *reinterpret_cast<const u16x8*>(tile + act_off(kb, mt * 16 + fi, fg));
}

// A-operand fragment (weights) straight from the blocked weight layout (wblk_off: 16 bytes per lane, lanes in address order)
__device__ __forceinline__ u16x8 w_frag(const u16* __restrict__ W, int NR, int kb, int nrow, int fg) {
  return *reinterpret_cast<const u16x8*>(W + wblk_off(NR, nrow, kb * 32 + fg * 8));
}

// the same fragment addressed as (16-row block, lane): with a wave-uniform block index (rows `nrow16`, a multiple of 16, and `kb` in
// SGPRs) the load is `global_load_dwordx4 v, v_lane16, s[base]` -- ONE address VGPR for all fragments of a phase
__device__ __forceinline__ u16x8 w_frag_u(const u16* __restrict__ W, int NR, int kb, int nrow16, int lane) {
  const long long blk = ((long long)kb * ((NR + 15) & ~15) + nrow16) * 32;
  return *reinterpret_cast<const u16x8*>(W + blk + lane * 8);
}

// store 4 consecutive channels (D layout: n = 16*nt16 + 4*fg + r) of token `row` into an activation tile
template <bool BF>
__device__ __forceinline__ void act_store4(u16* tile, int n, int row, f32x4 v) {
  const int kb = n >> 5, k = n & 31;
  *reinterpret_cast<u16x4*>(tile + act_off(kb, row, k >> 3) + (k & 7)) = cvt4<BF>(v);
}

// `rot` rotates the order in which the k-blocks are visited (register slot j holds k-block (j + rot) % KB).  Workgroups
// use different rotations so that the CUs of an XCD do not all stream the same weight lines from the same L2 channel at
// the same moment.
template <bool BF, int NT, int MT, int KB>
__device__ __forceinline__ void gemm_phase(f32x4 (&acc)[NT][MT], const u16x8 (&wf)[NT][KB], const u16* tile, int fi, int fg,
                                           int rot = 0) {
#pragma unroll
  for (int j = 0; j < KB; ++j) {
    const int kb = (j + rot) & (KB - 1);
    u16x8 b[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) b[mt] = act_frag(tile, kb, mt, fi, fg);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = H16<BF>::mfma(wf[nt][j], b[mt], acc[nt][mt]);
  }
}

// Same, but while computing with `wf` it also issues the loads of the NEXT phase's fragments, NT per k-step, so the
// 1-KiB weight loads are spread between the MFMAs instead of arriving as one burst that backs up the address path.
struct NoKStamps {};
struct KStamps8 { unsigned long long t[8]; };
template <bool BF, int NT, int MT, int KB, int NTN, class KS = NoKStamps>
__device__ __forceinline__ void gemm_phase_pf(f32x4 (&acc)[NT][MT], const u16x8 (&wf)[NT][KB], const u16* tile, int fi, int fg,
                                              int rot, u16x8 (&wnext)[NTN][KB], const u16* __restrict__ Wn, int NRn, int kb0n,
                                              int nrow0n, int rotn, [[maybe_unused]] KS* ks = nullptr /* diagnostic: per-k-step stamps */) {
  u16x8 bcur[MT], bnxt[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) bcur[mt] = act_frag(tile, rot & (KB - 1), mt, fi, fg);
#pragma unroll
  for (int j = 0; j < KB; ++j) {
    if (j + 1 < KB) {
      const int kb = (j + 1 + rot) & (KB - 1);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) bnxt[mt] = act_frag(tile, kb, mt, fi, fg);
    }
#pragma unroll
    for (int nt = 0; nt < NTN; ++nt) wnext[nt][j] = w_frag(Wn, NRn, kb0n + ((j + rotn) & (KB - 1)), nrow0n + nt * 16 + fi, fg);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = H16<BF>::mfma(wf[nt][j], bcur[mt], acc[nt][mt]);
    // one k-step of B fragments ahead, never more: keeps the LDS reads in flight far below the 4-bit lgkmcnt limit
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!std::is_same<KS, NoKStamps>::value) {
      if (ks) {
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ks->t[j])::"memory");
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) bcur[mt] = bnxt[mt];
  }
}


// ---- row-wise epilogue ---------------------------------------------------------------------------------------------
// The MFMA result layout gives a lane 4 consecutive channels of 16 different token rows: touching fp32 [rows][256]
// tensors that way means 64-byte column slices at a 1-KiB stride (measured: ~20k cycles just to issue 8 such loads per
// lane).  Instead the accumulators go to an LDS fp32 tile [64][kEpiLd] and every wave then handles whole rows: one
// 1-KiB coalesced access per wave instruction for the residual read and for the output store.
constexpr int kEpiLd = 256 + 4;   // floats per LDS row (+16 B pad: the 16 lanes of a D-layout write hit distinct banks)

__device__ __forceinline__ void epi_put(float* tile, int row, int n, f32x4 v) {
  *reinterpret_cast<float4*>(tile + row * kEpiLd + n) = float4{v[0], v[1], v[2], v[3]};
}

template <int NT, int KB>
__device__ __forceinline__ void load_wfrags(u16x8 (&wf)[NT][KB], const u16* __restrict__ W, int NR, int kb0, int nrow0, int fi, int fg,
                                            int rot = 0) {
#pragma unroll
  for (int j = 0; j < KB; ++j)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) wf[nt][j] = w_frag(W, NR, kb0 + ((j + rot) & (KB - 1)), nrow0 + nt * 16 + fi, fg);
}

// =====================================================================================================
// norm1 -> FFN -> norm2, C = 256 (WC/temporal_attention.py:181-185, :217-218), as a workgroup-level body that both the
// stand-alone kernel and the fused trajectory+FFN kernel run.  The 64 input rows sit in LDS as fp32 (`xtile`, row stride
// kEpiLd) and never return to HBM: norm1 overwrites them with y (the FFN residual), the linear2 accumulators are added in
// place, norm2 is applied per whole row on the way out.
// LDS: xtile 65 KiB fp32 | ytile 32 KiB | htile 32 KiB (single buffer: 2 barriers per 256-unit chunk) | parameters.
// =====================================================================================================
// diagnostic builds (-DAXVS_STAMPS -DAXVS_STAMPS_FFN): phase stamps INSIDE ffn_body instead of the enclosing kernel's
#if defined(AXVS_STAMPS) && defined(AXVS_STAMPS_FFN)
#define FSTAMP_DECL unsigned long long fst_[16] = {}
#define FSTAMP(s)                                                                          \
  do {                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                     \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(fst_[s])::"memory");      \
    __builtin_amdgcn_sched_barrier(0);                                                     \
  } while (0)
#define FSTAMP_FLUSH(n)                                                                    \
  do {                                                                                     \
    if (blockIdx.x < 8 && (threadIdx.x & 63) == 0)                                         \
      for (int i_ = 0; i_ < (n); ++i_) ::axvs::g_stamps[i_ * 64 + (blockIdx.x & 7) * 8 + (threadIdx.x >> 6)] = fst_[i_]; \
  } while (0)
#else
#define FSTAMP_DECL
#define FSTAMP(s)
#define FSTAMP_FLUSH(n)
#endif

// flag bits of the `wt` argument of ffn_body / the kernels that carry it: bit 0 = write-through (sc1) output rows; bits 4, 5 = the
// output rows are 16-bit (f16 / bf16) instead of fp32 (`out` then points at a [rows][256] 16-bit map)
constexpr int kOutF16 = 16, kOutBf16 = 32, kOut16Mask = kOutF16 | kOutBf16;

struct FfnLds {
  float* xtile;   // [64][kEpiLd] fp32
  u16* ytile;     // [8][64][32]
  u16* htile;     // [8][64][32]
  float* par;     // b1[F] | b2 | g1 | be1 | g2 | be2  (256 each)
};

struct FfnArgs {   // kernel argument block of the FFN half (packed weights, fp32 parameters)
  const u16 *W1, *W2;
  const float *b1, *b2, *g1, *be1, *g2, *be2;
  int F;
};

__device__ __forceinline__ void ffn_stage_params(const FfnLds& l, const float* __restrict__ b1, const float* __restrict__ b2,
                                                 const float* __restrict__ g1, const float* __restrict__ be1,
                                                 const float* __restrict__ g2, const float* __restrict__ be2, int F, int tid) {
  for (int i = tid; i < F; i += 512) {
    l.par[i] = b1[i];
    lds_fence();                       // F / 512 writes: keep lgkmcnt bounded for any d_ffn
  }
  if (tid < 256) {
    float* q = l.par + F;
    q[tid] = b2[tid];
    q[256 + tid] = g1[tid];
    q[512 + tid] = be1[tid];
    q[768 + tid] = g2[tid];
    q[1024 + tid] = be2[tid];
  }
}

// Precondition: xtile rows 8*wave .. 8*wave+7 were written by THIS wave (or a barrier has passed), parameters staged and a
// barrier passed since; w1f holds the linear1 fragments of chunk `crot` (rotation `rot`).
// `row_off(r)`: element offset of tile row r in `out`, or < 0 for a row that does not exist (ragged last tile).
// GELU: the hidden activation is the exact (erf) GELU of F.gelu instead of ReLU (WC/temporal_attention.py:9-17) -- a template flag so
// that the ReLU kernels (every shipped config, and the width-pass kernel that carries this body) keep their code as it is.
// PRE / `pre`: the wave's 8 input rows (row 8 wave + rr, float4 column `lane`) handed over in registers -- the caller then
// has NOT written them to xtile, and norm1 below skips reading them back (the fused width-pass kernel: its residual epilogue and
// this norm1 walk the same rows; one LDS round trip per row less).
struct NoRows { float4 v[8]; };
template <bool BF, class RowOff, bool GELU = false, bool PRE = false>
__device__ __forceinline__ void ffn_body(const FfnLds& l, u16x8 (&w1f)[2][8], const u16* __restrict__ W1, const u16* __restrict__ W2,
                                         float* __restrict__ out, RowOff row_off, int F, int rot, int crot, int tid, int wt = 0,
                                         const NoRows& pre = NoRows{}) {
  constexpr int C = 256, KB = 8;
  const int lane = tid & 63, wave = tid >> 6, fi = lane & 15, fg = lane >> 4;
  const float* sb1 = l.par;
  const float* sb2 = l.par + F;
  const float *sg1 = sb2 + C, *sbe1 = sb2 + 2 * C, *sg2 = sb2 + 3 * C, *sbe2 = sb2 + 4 * C;
  const int nchunk = F / 256;
  u16x8 w2f[2][KB];
  FSTAMP_DECL;
  FSTAMP(0);

  // ---- norm1, row-wise: y (fp32) back into xtile, y (16-bit) into ytile ----
  {
    const float4 gg = *reinterpret_cast<const float4*>(sg1 + lane * 4), bb = *reinterpret_cast<const float4*>(sbe1 + lane * 4);
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int r = wave * 8 + rr;
      float4 v;
      if constexpr (PRE) v = pre.v[rr];
      else v = *reinterpret_cast<const float4*>(l.xtile + r * kEpiLd + lane * 4);
      const float mu = wave_sum(v.x + v.y + v.z + v.w) * (1.f / C);
      const float a = v.x - mu, b = v.y - mu, c = v.z - mu, d = v.w - mu;
      const float rstd = rsqrtf(wave_sum(a * a + b * b + c * c + d * d) * (1.f / C) + 1e-5f);
      const f32x4 y = {a * rstd * gg.x + bb.x, b * rstd * gg.y + bb.y, c * rstd * gg.z + bb.z, d * rstd * gg.w + bb.w};
      *reinterpret_cast<float4*>(l.xtile + r * kEpiLd + lane * 4) = float4{y[0], y[1], y[2], y[3]};
      act_store4<BF>(l.ytile, lane * 4, r, y);
      if (rr == 3) lds_fence();
    }
  }
  FSTAMP(1);
  __syncthreads();
  FSTAMP(2);

  f32x4 acc2[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc2[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

#if defined(AXVS_STAMPS) && defined(AXVS_STAMPS_FFNK)
  KStamps8 kst1{}, kst2{};
#endif
#if defined(AXVS_STAMPS) && defined(AXVS_STAMPS_FFNC)     // unconditional stamps at 7 points of EVERY chunk (no loop unswitching): the last two chunks survive
  KStamps8 ccur{}, cprev{};
#define CSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ccur.t[i])::"memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define CSTAMP(i)
#endif
  for (int ci = 0; ci < nchunk; ++ci) {
#if defined(AXVS_STAMPS) && defined(AXVS_STAMPS_FFNC)
    cprev = ccur;
#endif
    CSTAMP(0);
    const int c = (ci + crot) % nchunk;                 // hidden-unit chunk handled in this iteration
#if defined(AXVS_STAMPS) && defined(AXVS_STAMPS_FFNK)
    if (ci == AXVS_STAMPS_FFNK) { FSTAMP(14); }
#endif
    const int cn = (min(ci + 1, nchunk - 1) + crot) % nchunk;
    // ---- linear1 + ReLU: my 32 hidden units of the chunk, all 64 rows; meanwhile fetch this chunk's linear2 fragments ----
    f32x4 acc1[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc1[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#if defined(AXVS_STAMPS) && defined(AXVS_STAMPS_FFNK)
    gemm_phase_pf<BF, 2, 4, KB, 2, KStamps8>(acc1, w1f, l.ytile, fi, fg, rot, w2f, W2, C, c * 8, wave * 32, rot, ci == AXVS_STAMPS_FFNK ? &kst1 : nullptr);
#else
    gemm_phase_pf<BF, 2, 4, KB, 2>(acc1, w1f, l.ytile, fi, fg, rot, w2f, W2, C, c * 8, wave * 32, rot);
#endif
    if (ci == 0) FSTAMP(3);
    CSTAMP(1);
    if (ci > 0) __syncthreads();                         // every wave is done reading the previous chunk's h
    CSTAMP(2);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int hn = c * 256 + wave * 32 + nt * 16 + fg * 4;          // global hidden index
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
        act_store4<BF>(l.htile, wave * 32 + nt * 16 + fg * 4, mt * 16 + fi, v);   // chunk-local hidden index
      }
    }
    if (ci == 0) FSTAMP(4);
    CSTAMP(3);
    __syncthreads();
    CSTAMP(4);
    if (ci == 0) FSTAMP(5);
    // ---- linear2 partial: W2[my 32 channels, chunk] . h, accumulated from zero and then added to the running sum (so the
    //      result is the ordered sum of per-chunk partials: exactly what the chunk-per-workgroup variant for few rows,
    //      ffn_split_kernel + ffn_finish_kernel, produces -- the choice between them depends on the row count and must not
    //      change a single bit); meanwhile fetch the next chunk's linear1 fragments (unconditional: the last iteration
    //      re-loads its own chunk, which keeps the vmcnt bookkeeping branch-free) ----
    f32x4 part[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) part[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#if defined(AXVS_STAMPS) && defined(AXVS_STAMPS_FFNK)
    if (ci == AXVS_STAMPS_FFNK) { FSTAMP(13); }
    gemm_phase_pf<BF, 2, 4, KB, 2, KStamps8>(part, w2f, l.htile, fi, fg, rot, w1f, W1, F, 0, cn * 256 + wave * 32, rot, ci == AXVS_STAMPS_FFNK ? &kst2 : nullptr);
#else
    gemm_phase_pf<BF, 2, 4, KB, 2>(part, w2f, l.htile, fi, fg, rot, w1f, W1, F, 0, cn * 256 + wave * 32, rot);
#endif
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc2[a][b] = ci == 0 ? part[a][b] : acc2[a][b] + part[a][b];
    CSTAMP(5);
    if (ci == 0) FSTAMP(6);
    if (ci == 1) FSTAMP(7);
    if (ci == 2) FSTAMP(8);
  }
  FSTAMP(9);

  // ---- xtile (= y) += acc2 + b2, in the accumulator layout; then norm2 per whole row and one 1-KiB store per row ----
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int n = wave * 32 + nt * 16 + fg * 4;
    const float4 b = *reinterpret_cast<const float4*>(sb2 + n);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      float* p = l.xtile + (mt * 16 + fi) * kEpiLd + n;
      const float4 y = *reinterpret_cast<const float4*>(p);
      *reinterpret_cast<float4*>(p) = float4{y.x + acc2[nt][mt][0] + b.x, y.y + acc2[nt][mt][1] + b.y, y.z + acc2[nt][mt][2] + b.z,
                                             y.w + acc2[nt][mt][3] + b.w};
    }
    lds_fence();
  }
  FSTAMP(10);
  __syncthreads();
  FSTAMP(11);
  {
    const float4 g2v = *reinterpret_cast<const float4*>(sg2 + lane * 4), be2v = *reinterpret_cast<const float4*>(sbe2 + lane * 4);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = wave * 8 + i;
      const float4 v = *reinterpret_cast<const float4*>(l.xtile + r * kEpiLd + lane * 4);
      const float mu = wave_sum(v.x + v.y + v.z + v.w) * (1.f / C);
      const float d0 = v.x - mu, d1 = v.y - mu, d2 = v.z - mu, d3 = v.w - mu;
      const float rstd = rsqrtf(wave_sum(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3) * (1.f / C) + 1e-5f);
      const long long off = row_off(r);
      if (off >= 0) {
        const float4 y = float4{d0 * rstd * g2v.x + be2v.x, d1 * rstd * g2v.y + be2v.y, d2 * rstd * g2v.z + be2v.z, d3 * rstd * g2v.w + be2v.w};
        if (wt & kOut16Mask) {      // the layer's output map in 16 bits (option "layer_out_dtype"): rows of 512 B, written here instead of by a cast pass
          const f32x4 yv = {y.x, y.y, y.z, y.w};
          const u16x4 h = (wt & kOutBf16) ? cvt4<true>(yv) : cvt4<false>(yv);
          u16* o16 = reinterpret_cast<u16*>(out);
          if (wt & 1) WtBuf(o16).store8((unsigned)((off + lane * 4) * 2), h);
          else *reinterpret_cast<u16x4*>(o16 + off + lane * 4) = h;
        } else if (wt & 1) WtBuf(out).store16((unsigned)((off + lane * 4) * 4), y);       // the next kernel reads these rows from other XCDs
        else *reinterpret_cast<float4*>(out + off + lane * 4) = y;
      }
      if (i == 3) lds_fence();
    }
  }
  FSTAMP(12);
  FSTAMP_FLUSH(16);
#if defined(AXVS_STAMPS) && defined(AXVS_STAMPS_FFNK)
  if (blockIdx.x < 8 && (threadIdx.x & 63) == 0)
    for (int i_ = 0; i_ < 8; ++i_) {
      ::axvs::g_stamps[(16 + i_) * 64 + (blockIdx.x & 7) * 8 + (threadIdx.x >> 6)] = kst1.t[i_];
      ::axvs::g_stamps[(24 + i_) * 64 + (blockIdx.x & 7) * 8 + (threadIdx.x >> 6)] = kst2.t[i_];
    }
#endif
#if defined(AXVS_STAMPS) && defined(AXVS_STAMPS_FFNC)
  if (blockIdx.x < 8 && (threadIdx.x & 63) == 0)
    for (int i_ = 0; i_ < 8; ++i_) {
      ::axvs::g_stamps[(32 + i_) * 64 + (blockIdx.x & 7) * 8 + (threadIdx.x >> 6)] = cprev.t[i_];
      ::axvs::g_stamps[(40 + i_) * 64 + (blockIdx.x & 7) * 8 + (threadIdx.x >> 6)] = ccur.t[i_];
    }
#endif
}

// stand-alone kernel: X fp32 [M][256] rows in, out rows out
template <bool BF, bool GELU = false>
__global__ __launch_bounds__(512) void ffn_fused_kernel(const float* __restrict__ X, const u16* __restrict__ W1,
                                                        const float* __restrict__ b1, const u16* __restrict__ W2,
                                                        const float* __restrict__ b2, const float* __restrict__ g1,
                                                        const float* __restrict__ be1, const float* __restrict__ g2,
                                                        const float* __restrict__ be2, float* __restrict__ out,
                                                        long long M, int F, RowStride rs, int oflags = 0 /* kOutF16 / kOutBf16 */) {
  constexpr int C = 256, KB = 8;
  extern __shared__ __attribute__((aligned(16))) char smem_c[];
  FfnLds l;
  l.xtile = reinterpret_cast<float*>(smem_c);
  l.ytile = reinterpret_cast<u16*>(smem_c + (size_t)kRows * kEpiLd * sizeof(float));
  l.htile = l.ytile + KB * kTileElems;
  l.par = reinterpret_cast<float*>(l.htile + KB * kTileElems);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fi = lane & 15, fg = lane >> 4;
  const long long m0 = (long long)blockIdx.x * kRows;
  const int nchunk = F / 256;
  // Every workgroup visits the k-blocks and hidden-unit chunks in the same order.  (Round 1 rotated both by workgroup index to
  // spread the weight stream over L2 channels: measured no gain, and a summation order that depends on the tile index made the
  // result of a clip depend on its position in the batch -- sharding a batch must be bit-exact.)
  constexpr int rot = 0, crot = 0;
  u16x8 w1f[2][KB];
  load_wfrags<2, KB>(w1f, W1, F, 0, crot * 256 + wave * 32, fi, fg, rot);
  ffn_stage_params(l, b1, b2, g1, be1, g2, be2, F, tid);
  {
    float4 rows[8];                      // my 8 rows, whole (1 KiB per wave instruction), all in flight together
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const long long m = min(m0 + wave * 8 + rr, M - 1);
      rows[rr] = *reinterpret_cast<const float4*>(X + rs.row(m) * C + lane * 4);
    }
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      *reinterpret_cast<float4*>(l.xtile + (wave * 8 + rr) * kEpiLd + lane * 4) = rows[rr];
      if (rr == 3) lds_fence();
    }
  }
  __syncthreads();                       // parameters staged
  auto row_off = [=](int r) { return m0 + r < M ? rs.row(m0 + r) * C : -1ll; };
  ffn_body<BF, decltype(row_off), GELU>(l, w1f, W1, W2, out, row_off, F, rot, crot, tid, oflags);
}

constexpr size_t kFfnTiles = (size_t)kRows * kEpiLd * sizeof(float) + 2 * 8 * kTileElems * sizeof(u16);   // x | y | h
inline size_t ffn_lds_bytes(int F) { return kFfnTiles + (size_t)(F + 5 * 256) * sizeof(float); }



}  // namespace axvs

namespace axvs {

// =====================================================================================================
// Temporal half of trajectory attention + output projection + residual, C = 256, 8 heads (d = 32).
//   WC/temporal_attention.py:60-75:  q2 = scale * proj_q(x[own frame]);  k2, v2 = proj_kv(x[frame f]) for every f;
//   a = softmax_f(q2 . k2);  o = sum_f a_f v2_f;  out = proj(o) (+ residual).
//
// Workgroup = MT*16 queries (sequence-order rows m'), 8 waves; wave w owns head w for the q2/k2/v2 projections and output
// channels 32w..32w+31 for the final projection.  The T-expanded tile x[f][k-block][row][32] sits in LDS (T * MT * 8 KiB);
// k2 and v2 never exist outside accumulators:
//   pass 1 (per frame): k2_f = Wk2_h x_f  ->  logit_f = q2 . k2_f   (k2 bias drops out of the softmax over f)
//   pass 2 (per frame): v2_f = Wv2_h x_f  ->  o += a_f * v2_f        (v2 bias added once: sum_f a_f = 1)
// Each wave streams exactly its own weight rows L2 -> VGPR (Wpq_h, Wk2_h, Wv2_h, Wp[32w..]: 64 KiB), one 64-VGPR fragment
// set that is refilled in place, slot by slot, right after a slot's last use.
// =====================================================================================================
template <int T, int MT, bool FFN = false>
constexpr size_t temporal_tile_bytes() {
  // x tile, re-used as: o tile | fp32 epilogue tile [| h tile of the FFN half; its y tile takes the o tile's place]
  size_t xt = (size_t)T * 8 * MT * 16 * 32 * sizeof(u16);
  size_t epi = (size_t)8 * MT * 16 * 32 * sizeof(u16) + (size_t)MT * 16 * kEpiLd * sizeof(float);
  if (FFN) epi += (size_t)8 * MT * 16 * 32 * sizeof(u16);
  return xt > epi ? xt : epi;
}
template <int T, int MT, bool FFN = false, int MQ = 0>
constexpr size_t temporal_lds_bytes(int F = 0) {   // tiles | bpq, bv2, bp | FFN parameters | MQ: bq, bk, bv
  return temporal_tile_bytes<T, MT, FFN>() + 3 * 256 * sizeof(float) + (FFN ? (size_t)(F + 5 * 256) * sizeof(float) : 0) +
         (MQ ? 3 * 256 * sizeof(float) : 0);
}

// MQ ("merged q/k/v"): the trajectory kernel of a pass computes q, k, v of its OWN 64 rows first (the body of qkv_fused_kernel:
// rows gathered through the RowMap, operand tiles in the not yet used x-tile space), keeps q in registers, stores K / V^T
// write-through, tells the other row tiles of its sequence through `sync[sequence]` and reads THEIR K / V^T with sc1 loads once
// all of them have arrived -- one launch per pass instead of two, q never leaves the CU (WC/temporal_attention.py:42-57,197-213).
struct OwnQkv {
  const float* src;       // fp32 token rows, addressed through the kernel's RowMap
  const float* pos;       // nullable; read when pg.mode == 0
  PosGen pg;
  const u16 *Wq, *Wk, *Wv;
  const float *bq, *bk, *bv;
  float qscale;           // head_dim^-0.5 * log2(e)
  unsigned* sync;         // [sequences] arrival counters: zero before the launch, zero again after it (see the kernel)
  int* status;            // nullable: bit 0 <- an operand left the fp16 range, bit 2 <- a hand-off wait ran out
  unsigned spin_limit;    // polls (with s_sleep) before a hand-off wait gives up (kSyncSpinLimit; option "sync_spin_limit")
};
constexpr unsigned kSyncSpinLimit = 1u << 22;    // ~ 1 s

template <int MT>
__device__ __forceinline__ int xt_off(int f, int kb, int row, int c) {   // element offset in the x tile
  return ((f * 8 + kb) * (MT * 16) + row) * 32 + swz_chunk(row, c) * 8;
}

// one GEMM sweep over the 8 k-blocks: acc[nt][mt] += W[slot j] . B(kb=j);  B fragments come from `bbase[mt] + kb*kbstride`
// (per-lane LDS element offsets), one k-step of lookahead.  If REFILL, slot j is re-loaded from Wn right after its use.
template <bool BF, int MT, bool REFILL, bool SWAP = false>   // SWAP: D[token][channel] instead of D[channel][token]
__device__ __forceinline__ void sweep8(f32x4 (&acc)[2][MT], u16x8 (&wf)[2][8], const u16* xt, const int (&bbase)[MT], int kbstride,
                                       const u16* __restrict__ Wn, int NRn, int nrow0n, int fi, int fg) {
  // B fragments run LA k-steps ahead of the MFMAs (LDS latency ~ 2 steps of 4 MFMAs); at most (LA+1)*MT <= 12 reads in flight
  constexpr int LA = MT <= 2 ? 2 : 1;
  u16x8 b[LA + 1][MT];
#pragma unroll
  for (int l = 0; l < LA; ++l)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) b[l][mt] = *reinterpret_cast<const u16x8*>(xt + bbase[mt] + l * kbstride);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (j + LA < 8) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) b[(j + LA) % (LA + 1)][mt] = *reinterpret_cast<const u16x8*>(xt + bbase[mt] + (j + LA) * kbstride);
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        acc[nt][mt] = SWAP ? H16<BF>::mfma(b[j % (LA + 1)][mt], wf[nt][j], acc[nt][mt])
                           : H16<BF>::mfma(wf[nt][j], b[j % (LA + 1)][mt], acc[nt][mt]);
    if (REFILL) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) wf[nt][j] = w_frag(Wn, NRn, j, nrow0n + nt * 16 + fi, fg);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// NKS = 0: the x tile is staged from global memory (X16, written by spatial_attn_kernel).
// NKS > 0: the spatial half runs right here (32*NKS >= L keys per frame): wave w computes head w's
//          x[q, f, :] = softmax_l(q . k[f, l]) v[f, l, :] for the 64 queries and all T frames from K / V^T fragments loaded
//          straight from L2 (no LDS staging; V^T comes pre-blocked from qkv_fused_kernel) and writes it into the LDS x tile --
//          the T-expanded tensor never touches HBM.  Needs the tile inside one sequence (N % (16*MT) == 0), L % 16 == 0.
// FFN: the rows do not go back to HBM after the residual; norm1 -> FFN -> norm2 (ffn_body) runs on them right here and `out`
//      receives the layer output (MT = 4 only).
// MQ: 1 = the kernel computes q, k, v of its own rows first (OwnQkv; 64-row tiles, frames of a multiple of 16 keys); 2 = the same
//     for frames of exactly 64 keys, where a row tile IS one frame of its sequence: that frame's K / V^T fragments are the
//     accumulators of the k / v sweeps (same 16-bit values as the stored ones), so its QK^T / softmax / AV run first, from registers,
//     while the sibling tiles' K / V^T stores drain (frames are visited in the order own, own + 1, ... mod T; x-tile blocks are
//     independent, so the result does not depend on the order).
// (Rounds 3 - 5 also carried a row-form-V variant, a variant that emitted the next pass's q/k/v from the epilogue, and a persistent team grid for
//  merged launches beyond two rounds of the chip: all three bit-identical and measured slower -- DESIGN.md, profiles/r5_persistent_merged.txt -- removed in round 6.)
template <bool BF, int T, int MT, int NKS = 0, bool FFN = false, int MQ = 0>
__global__ __launch_bounds__(512) void temporal_fused_kernel(const u16* __restrict__ X16 /* [8][T][Mp][32] */,
                                                             const u16* __restrict__ Wpq_a, const float* __restrict__ bpq,
                                                             const u16* __restrict__ Wpkv_a, const float* __restrict__ bpkv,
                                                             const u16* __restrict__ Wp_a, const float* __restrict__ bp,
                                                             const float* __restrict__ res /* required */, float* __restrict__ out,
                                                             RowMap rm, long long Mp, int N_a, int L_a, float scale, const u16* __restrict__ Q16 = nullptr,
                                                             const u16* __restrict__ K16 = nullptr,
                                                             const u16* __restrict__ VT16 = nullptr, FfnArgs fa_a = FfnArgs{},
                                                             const u16* __restrict__ Wk2T_a = nullptr /* [8*256][32], pack_wk2t_kernel */,
                                                             int wt = 0 /* bit 0: write-through output rows (byte offsets < 4 GiB); FFN: bits 4 / 5 = 16-bit output map (kOutF16 / kOutBf16) */,
                                                             int spatial_only = 0 /* measurement: 1 = stop after the QK^T / AV half; MQ kernels also 2 = stop after their q/k/v part */,
                                                             const float* __restrict__ ln_g = nullptr /* post-norm LayerNorm(x + attn) */,
                                                             const float* __restrict__ ln_b = nullptr,
                                                             OwnQkv oq_a = OwnQkv{}) {
  static_assert(!FFN || MT == 4, "the FFN half works on 64-row tiles");
  static_assert(MQ == 0 || ((MT == 4 || MT == 2 || MT == 1) && NKS > 0), "own q/k/v: 64-, 32- or 16-row tiles");
  static_assert(MQ != 2 || (MT == 4 && NKS == 2 && T >= 2), "own frame first: a 64-row tile is one 64-key frame");
  constexpr int C = 256, ROWS = MT * 16;
  extern __shared__ __attribute__((aligned(16))) u16 smem[];
  u16* xt = smem;                                              // [T][8][ROWS][32]; later re-used as the o tile [8][ROWS][32]
  float* sbias = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + temporal_tile_bytes<T, MT, FFN>());   // bpq | bv2 | bp
  float* const sqkvb = sbias + 3 * C + (FFN ? fa_a.F + 5 * C : 0);   // MQ: bq | bk | bv
  FfnLds fl;
  if constexpr (FFN) {
    fl.ytile = xt;
    fl.xtile = reinterpret_cast<float*>(xt + 8 * ROWS * 32);
    fl.htile = reinterpret_cast<u16*>(fl.xtile + ROWS * kEpiLd);
    fl.par = sbias + 3 * C;
  }

  // XCD-aware tile order (speed only): blocks b and b+8 share an XCD, hence an L2.  With the spatial half in the kernel the
  // N/ROWS row tiles of one sequence all read that sequence's K / V -- put them on the same XCD so it is fetched once.
  // Row tiles: with the spatial half in the kernel a tile never straddles two sequences (its queries share one K / V): every
  // sequence gets ceil(N / ROWS) tiles, the last one partly filled -- any axis length works, rows past the sequence's end are
  // clamped copies that are computed and never stored.  Without it (x staged from HBM) tiles are plain ROWS-row slices.
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fi = lane & 15, fg = lane >> 4;
  const int N = N_a, L = L_a;
  const u16* const Wpq = Wpq_a;
  const u16* const Wpkv = Wpkv_a;
  const u16* const Wp = Wp_a;
  const u16* const Wk2T = Wk2T_a;
  const FfnArgs& fa = fa_a;
  const OwnQkv& oq = oq_a;
  long long tile = blockIdx.x;
  long long m0;                                                 // first sequence-order row of the tile
  int nvalid;                                                   // rows of the tile that exist
  if constexpr (NKS > 0) {
    const int tps = (N + ROWS - 1) / ROWS;                      // tiles per sequence
    if (gridDim.x % (8 * tps) == 0) {
      const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
      tile = ((long long)(j / tps) * 8 + xcd) * tps + (j % tps);
    }
    const long long sq = tile / tps;
    const int n0 = (int)(tile - sq * tps) * ROWS;
    m0 = sq * N + n0;
    nvalid = min(ROWS, N - n0);
  } else {
    m0 = tile * ROWS;
    nvalid = (int)min((long long)ROWS, Mp - m0);
  }

  // padded frames (RowMap): L is the frame length of the row space (a multiple of 16), Lr the number of keys / rows that exist
  [[maybe_unused]] const int Lr = rm.Lv ? rm.Lv : L;
  AXVS_STAMP_DECL;
  AXVS_WG_BEGIN;
  AXVS_STAMP(0);
  // biases (and the FFN half's parameters) -> LDS
  auto stage_small = [&]() {
    if (tid < C) {
      sbias[tid] = bpq[tid];
      sbias[C + tid] = bpkv[C + tid];       // v2 half of the proj_kv bias
      sbias[2 * C + tid] = bp[tid];
    }
    if constexpr (FFN) {
      lds_fence();
      ffn_stage_params(fl, fa.b1, fa.b2, fa.g1, fa.be1, fa.g2, fa.be2, fa.F, tid);
    }
  };
  // first weight set: Wpq rows of my head (with a long in-kernel attention phase it is fetched after that phase instead,
  // to keep 64 VGPRs free for the score tiles)
  u16x8 wf[2][8];
  if constexpr (NKS == 0) load_wfrags<2, 8>(wf, Wpq, C, 0, wave * 32, fi, fg);

  if constexpr (NKS > 0) {
    // ---- spatial half, head = wave (WC/temporal_attention.py:46-57) ----
    const long long seq0 = (m0 / N) * N;                        // first row of my sequence
    const u16* Qh = Q16 + (long long)wave * Mp * 32;
    const u16* Kh = K16 + (long long)wave * Mp * 32;
    const long long nsf = Mp / L;                               // frame slots (sequences x frames)
    const u16* Vh = VT16 + (long long)wave * nsf * NKS * 1024;
    u16x8 qf[MT];
    // K fragments run one frame ahead, V^T fragments are requested at the top of their frame (before the scores):
    // the L2 latency of both hides behind MFMA + softmax work
    u16x8 kb[2][2 * NKS], vf[2][NKS];                         // K fragments: two sets used alternately (frame parity), no copies
    // MQ: K / V^T written by the sibling tiles of this launch are read with sc1 buffer loads (per-lane byte offset + a
    // wave-uniform frame offset), and only after the hand-off
    [[maybe_unused]] const ScBuf kbuf(K16), vbuf(VT16);
    [[maybe_unused]] unsigned kvo[2 * NKS], vvo = 0;
    [[maybe_unused]] const unsigned kstep_b = (unsigned)L * 64u, vstep_b = NKS * 2048u;
    [[maybe_unused]] unsigned* const cnt = MQ != 0 ? oq.sync + m0 / N : nullptr;      // my sequence's arrival counter
    [[maybe_unused]] const unsigned tps_u = (unsigned)((N + ROWS - 1) / ROWS);
    [[maybe_unused]] u16x8 vown[2][NKS];          // MQ == 2: V^T fragments of my own frame
    // A wait that runs out (a sibling tile that is not co-resident: another tenant, a profiler) must never turn into plausible numbers: the wave
    // reports it (status bit 2) and POISONS every softmax it computes from then on (poison = NaN multiplies the reciprocal row sum; 1.0 otherwise:
    // exact), so the rows of this tile come out NaN instead of being computed from stale K / V^T (round 6; rounds 4 - 5 computed on).
    [[maybe_unused]] float poison = 1.f;
    auto wait_siblings = [&]() {            // every wave polls for itself (one dword, sc1) and loads only after its poll matched
      unsigned spins = 0;
      while (ld_sc1_u32(cnt) < tps_u) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > oq.spin_limit) {
          if (oq.status != nullptr && lane == 0) atomicOr(oq.status, 4);
          poison = __uint_as_float(__builtin_amdgcn_readfirstlane(0x7fc00000u));      // (kept in a scalar register)
          break;
        }
      }
    };
    if constexpr (MQ != 0) {
      // ---- q, k, v of my own 64 rows (WC/temporal_attention.py:42-44 with query = key = src + pos, value = src, :200-203): the body
      //      of qkv_fused_kernel on this tile's rows.  The operand tiles live in the x-tile space, which nothing uses yet.
      //      Order: v, k -- their stores are what the sibling tiles wait for -- then the hand-off signal, then q (which never
      //      leaves the CU): the q sweep and, for MQ == 2, the own frame's scores run while the signals travel. ----
      constexpr int KBSq = ROWS * 32;
      u16* const tqk = xt;                       // (src + pos) tile [8][64][32]
      u16* const tv = xt + 8 * KBSq;             // src tile
      // every small parameter and the first weight fragments are requested once and parked (LDS / registers) behind the row gather
      float sm3[3], sq3[3], sf5[5], sf1[2];
      auto request_params = [&]() {
        load_wfrags<2, 8>(wf, oq.Wv, C, 0, wave * 32, fi, fg);
        if (tid < C) {
          sq3[0] = oq.bq[tid]; sq3[1] = oq.bk[tid]; sq3[2] = oq.bv[tid];
          sm3[0] = bpq[tid]; sm3[1] = bpkv[C + tid]; sm3[2] = bp[tid];
          if constexpr (FFN) {
            sf5[0] = fa.b2[tid]; sf5[1] = fa.g1[tid]; sf5[2] = fa.be1[tid]; sf5[3] = fa.g2[tid]; sf5[4] = fa.be2[tid];
          }
        }
        if constexpr (FFN) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
            if (tid + i * 512 < fa.F) sf1[i] = fa.b1[tid + i * 512];      // (d_ffn > 1024: the rest follows the gather)
        }
      };
      {
        // thread -> (row, float4 column); RowMap arithmetic once per wave: lane k computes the k-th of the wave's ROWS / 8 rows (rows
        // wave + 8k; rows past the sequence's end are clamped copies), v_readlane broadcasts
        constexpr int RPWq = ROWS / 8;               // rows per wave: 8 (64-row tiles) or 2 (16-row tiles)
        constexpr int GRPq = RPWq >= 4 ? 4 : RPWq;   // rows of a wave in flight together
        int off_lo, off_hi, coords = 0;
        {
          const int myrow = wave + 8 * (lane & (RPWq - 1));
          const int mym = (int)m0 + min(myrow, nvalid - 1);
          const long long myoff = (oq.pg.mode ? nat_row_coords(rm, mym, oq.pg.l_is_h, &coords) : nat_row(rm, mym)) * C;
          off_lo = (int)(myoff & 0xffffffffll);
          off_hi = (int)(myoff >> 32);
        }
        PosGenLane pl;
        if (oq.pg.mode) pl.init(oq.pg, lane * 4);
        float amax = 0.f;
#pragma unroll
        for (int half = 0; half < RPWq / GRPq; ++half) {
          float4 a[GRPq], p[GRPq];
#pragma unroll
          for (int i = 0; i < GRPq; ++i) {
            const int k = half * GRPq + i;
            const long long off = (((long long)__builtin_amdgcn_readlane(off_hi, k) << 32) | (unsigned)__builtin_amdgcn_readlane(off_lo, k)) + lane * 4;
            a[i] = *reinterpret_cast<const float4*>(oq.src + off);
            if (oq.pg.mode) p[i] = pl.eval(oq.pg, __builtin_amdgcn_readlane(coords, k));      // sine embedding generated, not read
            else p[i] = oq.pos ? *reinterpret_cast<const float4*>(oq.pos + off) : float4{0.f, 0.f, 0.f, 0.f};
          }
          if (half == 0) request_params();   // behind the first group of row requests (-0.45 us on the width-pass kernel, A/B)
#pragma unroll
          for (int i = 0; i < GRPq; ++i) {
            const int row = wave + 8 * (half * GRPq + i);
            const int n = lane * 4, kbq = n >> 5, k = n & 31;
            const int o = (kbq * ROWS + row) * 32 + swz_chunk(row, k >> 3) * 8 + (k & 7);
            *reinterpret_cast<u16x4*>(tv + o) = cvt4<BF>(f32x4{a[i].x, a[i].y, a[i].z, a[i].w});
            *reinterpret_cast<u16x4*>(tqk + o) = cvt4<BF>(f32x4{a[i].x + p[i].x, a[i].y + p[i].y, a[i].z + p[i].z, a[i].w + p[i].w});
            if (!BF) {
              amax = fmaximum(amax, fmaximum(fmaximum(fabsf(a[i].x), fabsf(a[i].y)), fmaximum(fabsf(a[i].z), fabsf(a[i].w))));
              amax = fmaximum(amax, fmaximum(fmaximum(fabsf(a[i].x + p[i].x), fabsf(a[i].y + p[i].y)), fmaximum(fabsf(a[i].z + p[i].z), fabsf(a[i].w + p[i].w))));
            }
          }
          lds_fence();
        }
        if (!BF && oq.status != nullptr && !(amax <= 65504.f)) atomicOr(oq.status, 1);
      }
      if (tid < C) {
        sqkvb[tid] = sq3[0]; sqkvb[C + tid] = sq3[1]; sqkvb[2 * C + tid] = sq3[2];
        sbias[tid] = sm3[0]; sbias[C + tid] = sm3[1]; sbias[2 * C + tid] = sm3[2];
        if constexpr (FFN) {
          float* q = fl.par + fa.F;
          q[tid] = sf5[0]; q[256 + tid] = sf5[1]; q[512 + tid] = sf5[2]; q[768 + tid] = sf5[3]; q[1024 + tid] = sf5[4];
        }
      }
      lds_fence();
      if constexpr (FFN) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
          if (tid + i * 512 < fa.F) fl.par[tid + i * 512] = sf1[i];
        lds_fence();
        for (int i = tid + 1024; i < fa.F; i += 512) {
          fl.par[i] = fa.b1[i];
          lds_fence();
        }
      }
      AXVS_STAMP(16);
      __syncthreads();
      AXVS_STAMP(17);
      int bb[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int row = mt * 16 + fi;
        bb[mt] = row * 32 + swz_chunk(row, fg) * 8;
      }
      const WtBuf wkb(K16), wvb(VT16);
      {   // v with the operands swapped (tokens on the D rows) -> block-transposed V^T, write-through
        f32x4 acc[2][MT];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < MT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        sweep8<BF, MT, true, true>(acc, wf, tv, bb, KBSq, oq.Wk, C, wave * 32, fi, fg);
        if (MT >= 2 && L % 32 == 0) {
          // tile pairs (mt, mt+1) are the two 16-key halves of one 32-key step: 16 contiguous bytes per lane
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            const float b = sqkvb[2 * C + wave * 32 + nt * 16 + fi];
#pragma unroll
            for (int mp = 0; mp + 1 < MT; mp += 2) {
              float v[8] = {acc[nt][mp][0] + b, acc[nt][mp][1] + b, acc[nt][mp][2] + b, acc[nt][mp][3] + b,
                            acc[nt][mp + 1][0] + b, acc[nt][mp + 1][1] + b, acc[nt][mp + 1][2] + b, acc[nt][mp + 1][3] + b};
              const u16x8 v8 = cvt8<BF>(v);
              if constexpr (MQ == 2) vown[nt][mp >> 1] = v8;     // my rows ARE the keys of my frame
              if (mp * 16 < nvalid) {
                const unsigned mt0 = (unsigned)m0 + mp * 16;
                const unsigned sf = mt0 / (unsigned)L;
                const int ks = (int)(mt0 - sf * L) >> 5;
                const long long d = ((((long long)wave * nsf + sf) * NKS + ks) * 2 + nt) * 512 + fi * 32 + fg * 8;
                wvb.store16((unsigned)(d * 2), v8);
              }
            }
          }
        } else {   // 16-key pieces: frames of an odd multiple of 16 keys (the host admits L % 16 == 0 only), and every 16-row tile
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            const float b = sqkvb[2 * C + wave * 32 + nt * 16 + fi];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              if (mt * 16 < nvalid) {
                const unsigned mt0 = (unsigned)m0 + mt * 16;
                const unsigned sf = mt0 / (unsigned)L;
                const int l = (int)(mt0 - sf * L) + fg * 4;
                const int ks = l >> 5, pp = fg * 8 + ((l >> 4) & 1) * 4;
                const long long d = ((((long long)wave * nsf + sf) * NKS + ks) * 2 + nt) * 512 + fi * 32 + pp;
                f32x4 v = acc[nt][mt];
                v[0] += b; v[1] += b; v[2] += b; v[3] += b;
                wvb.store8((unsigned)(d * 2), cvt4<BF>(v));
                // the frame's last 16-key tile also clears the padding half of its 32-key step (finite values for probability 0)
                if (L % 32 != 0 && (int)(mt0 - sf * L) + 16 == L) wvb.store8((unsigned)((d ^ 4) * 2), u16x4{0, 0, 0, 0});
              }
            }
          }
        }
      }
      AXVS_STAMP(18);
      {   // k: rows of K16 (perm32 channel order: 16 contiguous bytes per lane), write-through
        f32x4 acc[2][MT];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < MT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        sweep8<BF, MT, true>(acc, wf, tqk, bb, KBSq, oq.Wq, C, wave * 32, fi, fg);
        const float4 b0 = *reinterpret_cast<const float4*>(sqkvb + C + wave * 32 + fg * 4);
        const float4 b1 = *reinterpret_cast<const float4*>(sqkvb + C + wave * 32 + 16 + fg * 4);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          float v[8] = {acc[0][mt][0] + b0.x, acc[0][mt][1] + b0.y, acc[0][mt][2] + b0.z, acc[0][mt][3] + b0.w,
                        acc[1][mt][0] + b1.x, acc[1][mt][1] + b1.y, acc[1][mt][2] + b1.z, acc[1][mt][3] + b1.w};
          const u16x8 k8 = cvt8<BF>(v);
          if constexpr (MQ == 2) kb[0][mt] = k8;
          if (mt * 16 + fi < nvalid) wkb.store16((unsigned)((((long long)wave * Mp + m0 + mt * 16 + fi) * 32 + fg * 8) * 2), k8);
        }
      }
      AXVS_STAMP(19);
      // hand-off, producer side: my K / V^T stores are complete (every wave drains its own, then the barrier), ONE lane adds to the
      // sequence's counter.  (The counter wraps back to 0 with the last of the 2 * tiles arrivals + departures: zero again at the end.)
      vm_drain();
      AXVS_STAMP(20);
      __syncthreads();
      if (tid == 0) atomicInc(cnt, 2 * tps_u - 1);
      AXVS_STAMP(21);
      if (spatial_only == 2) {                   // measurement: stop behind the q/k/v part (bench.py subtracts this from spatial_only = 1)
        if (tid == 0) atomicInc(cnt, 2 * tps_u - 1);      // depart at once: nobody polls, the counter still ends at zero
        return;
      }
      {   // q: stays in registers as the B operand of QK^T (the values qkv_fused_kernel would store and this kernel load back)
        f32x4 acc[2][MT];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < MT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        sweep8<BF, MT, false>(acc, wf, tqk, bb, KBSq, oq.Wq, C, 0, fi, fg);
        const float4 b0 = *reinterpret_cast<const float4*>(sqkvb + wave * 32 + fg * 4);
        const float4 b1 = *reinterpret_cast<const float4*>(sqkvb + wave * 32 + 16 + fg * 4);
        const float sc_ = oq.qscale;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          float v[8] = {(acc[0][mt][0] + b0.x) * sc_, (acc[0][mt][1] + b0.y) * sc_, (acc[0][mt][2] + b0.z) * sc_, (acc[0][mt][3] + b0.w) * sc_,
                        (acc[1][mt][0] + b1.x) * sc_, (acc[1][mt][1] + b1.y) * sc_, (acc[1][mt][2] + b1.z) * sc_, (acc[1][mt][3] + b1.w) * sc_};
          qf[mt] = cvt8<BF>(v);
        }
      }
      AXVS_STAMP(22);
      __syncthreads();                           // every wave is done with the operand tiles: the x tile may be written
#pragma unroll
      for (int kt = 0; kt < 2 * NKS; ++kt)
        kvo[kt] = (unsigned)((((long long)wave * Mp + seq0 + min(kt * 16 + fi, Lr - 1)) * 32 + fg * 8) * 2);
      vvo = (unsigned)((((long long)wave * nsf + seq0 / L) * (NKS * 1024) + fi * 32 + fg * 8) * 2);
      if constexpr (MQ == 1) {
        wait_siblings();
        AXVS_STAMP(23);
#pragma unroll
        for (int kt = 0; kt < 2 * NKS; ++kt) kb[0][kt] = kbuf.load16(kvo[kt], 0);
      }
    } else {
#pragma unroll
    for (int qt = 0; qt < MT; ++qt) qf[qt] = *reinterpret_cast<const u16x8*>(Qh + (m0 + min(qt * 16 + fi, nvalid - 1)) * 32 + fg * 8);
    }
    const bool ragged = Lr != NKS * 32;
    u16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = H16<BF>::from_f32(1.f);
    // per-lane fragment pointers, advanced by one frame per iteration (all index math hoisted out of the loop:
    // frame slot sf = seq0 / L + f because N = T * L)
    const u16* kp[2 * NKS];
#pragma unroll
    for (int kt = 0; kt < 2 * NKS; ++kt) kp[kt] = Kh + (seq0 + min(kt * 16 + fi, Lr - 1)) * 32 + fg * 8;
    const u16* vp = Vh + (seq0 / L) * (NKS * 1024) + fi * 32 + fg * 8;
    const int kstep = L * 32;
    if constexpr (MQ == 0) {
#pragma unroll
    for (int kt = 0; kt < 2 * NKS; ++kt) kb[0][kt] = *reinterpret_cast<const u16x8*>(kp[kt]);
    }
    AXVS_STAMP(11);
    if constexpr (MQ == 0) stage_small();   // behind the cold Q / K loads of the first frame instead of in front of the barrier
    // One frame.  PAR = f & 1 as a compile-time constant (the K fragment sets alternate; register arrays need static indices).
    // LAST: the sequence's last frame -- no K prefetch for a next one, and the first GEMM phase's weight fragments (Wpq rows of
    // my head) are requested as soon as the score registers are free, so their L2 latency hides behind the last AV products and
    // the barrier instead of following them.
#ifdef AXVS_STAMPS
    int nvis_ = 0;
#endif
    // MQ: `fnext` is the frame visited after f (its K fragments are requested here).  OWN (MQ == 2, first frame): K / V^T fragments
    // of frame f are this tile's own rows (kb[PAR] / vown, from the sweeps); the wait for the sibling tiles sits right behind the
    // score MFMAs, and the K AND V^T fragments of frame fnext are requested there, so that they arrive behind the softmax and the
    // AV products of the own frame.  VPRE: the V^T fragments of frame f were requested by the frame before (the one after OWN).
    auto frame = [&](const int f, const int fnext, auto par_tag, auto last_tag, auto own_tag, auto vpre_tag) {
      constexpr int PAR = decltype(par_tag)::value;
      constexpr bool LAST = decltype(last_tag)::value;
      constexpr bool OWN = decltype(own_tag)::value;
      constexpr bool VPRE = decltype(vpre_tag)::value;
      if constexpr (MQ != 0) {
        if constexpr (!OWN) {
          if constexpr (!VPRE) {
#pragma unroll
            for (int nd = 0; nd < 2; ++nd)
#pragma unroll
              for (int ks = 0; ks < NKS; ++ks) vf[nd][ks] = vbuf.load16(vvo + (ks * 2 + nd) * 1024, (unsigned)f * vstep_b);
          }
          if constexpr (!LAST) {
#pragma unroll
            for (int kt = 0; kt < 2 * NKS; ++kt) kb[PAR ^ 1][kt] = kbuf.load16(kvo[kt], (unsigned)fnext * kstep_b);
          }
        }
      } else {
#pragma unroll
      for (int nd = 0; nd < 2; ++nd)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) vf[nd][ks] = *reinterpret_cast<const u16x8*>(vp + (ks * 2 + nd) * 512);
      vp += NKS * 1024;
      if constexpr (!LAST) {
#pragma unroll
        for (int kt = 0; kt < 2 * NKS; ++kt) kp[kt] += kstep;
#pragma unroll
        for (int kt = 0; kt < 2 * NKS; ++kt) kb[PAR ^ 1][kt] = *reinterpret_cast<const u16x8*>(kp[kt]);
      }
      }
      f32x4 sc[MT][2 * NKS];
#pragma unroll
      for (int kt = 0; kt < 2 * NKS; ++kt) {
#pragma unroll
        for (int qt = 0; qt < MT; ++qt) sc[qt][kt] = H16<BF>::mfma(kb[PAR][kt], qf[qt], f32x4{0.f, 0.f, 0.f, 0.f});   // D[key][query]
      }
      if constexpr (OWN) {                 // hand-off, consumer side: K / V^T of the next frame, from the sibling tiles
        AXVS_STAMP(23);
        wait_siblings();
#pragma unroll
        for (int kt = 0; kt < 2 * NKS; ++kt) kb[PAR ^ 1][kt] = kbuf.load16(kvo[kt], (unsigned)fnext * kstep_b);
#pragma unroll
        for (int nd = 0; nd < 2; ++nd)
#pragma unroll
          for (int ks = 0; ks < NKS; ++ks) vf[nd][ks] = vbuf.load16(vvo + (ks * 2 + nd) * 1024, (unsigned)fnext * vstep_b);
        AXVS_STAMP(11);
      }
      if (ragged) {
#pragma unroll
        for (int kt = 0; kt < 2 * NKS; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (kt * 16 + fg * 4 + r >= Lr) {
#pragma unroll
              for (int qt = 0; qt < MT; ++qt) sc[qt][kt][r] = -INFINITY;
            }
      }
      float inv[MT];
      u16x8 pf[MT][NKS];
#pragma unroll
      for (int qt = 0; qt < MT; ++qt) {
        float mx = sc[qt][0][0];
#pragma unroll
        for (int kt = 0; kt < 2 * NKS; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) mx = fmaximum(mx, sc[qt][kt][r]);      // v_maximum3_f32: no NaN-quieting copies
        mx = groups_maximum(mx);
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 mx2 = {mx, mx};
#pragma unroll
        for (int kt = 0; kt < 2 * NKS; ++kt) {
          const f32x2 lo = f32x2{sc[qt][kt][0], sc[qt][kt][1]} - mx2, hi = f32x2{sc[qt][kt][2], sc[qt][kt][3]} - mx2;   // v_pk_add_f32
          sc[qt][kt][0] = __builtin_amdgcn_exp2f(lo[0]); sc[qt][kt][1] = __builtin_amdgcn_exp2f(lo[1]);
          sc[qt][kt][2] = __builtin_amdgcn_exp2f(hi[0]); sc[qt][kt][3] = __builtin_amdgcn_exp2f(hi[1]);
        }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
          for (int j = 0; j < 8; ++j) pf[qt][ks][j] = H16<BF>::from_f32(sc[qt][2 * ks + (j >> 2)][j & 3]);
        // softmax denominator on the (otherwise idle) matrix pipe: ones . P^T sums the 32 keys of a step across all lanes,
        // so every D row already holds the full row sum of its query -- no VALU adds, no cross-lane reduction
        f32x4 ssum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) ssum = H16<BF>::mfma(ones, pf[qt][ks], ssum);
        inv[qt] = __builtin_amdgcn_rcpf(ssum[0]);   // 1 ulp; the result is rounded to 16 bits right after
        if constexpr (MQ != 0) inv[qt] *= poison;   // (x 1.0 unless a hand-off wait ran out)
      }
      if constexpr (LAST) load_wfrags<2, 8>(wf, Wpq, C, 0, wave * 32, fi, fg);
      f32x4 xa[MT][2];
#pragma unroll
      for (int nd = 0; nd < 2; ++nd) {
#pragma unroll
        for (int qt = 0; qt < MT; ++qt) xa[qt][nd] = H16<BF>::mfma(OWN ? vown[nd][0] : vf[nd][0], pf[qt][0], f32x4{0.f, 0.f, 0.f, 0.f});   // D[d][query]
#pragma unroll
        for (int ks = 1; ks < NKS; ++ks) {
#pragma unroll
          for (int qt = 0; qt < MT; ++qt) xa[qt][nd] = H16<BF>::mfma(OWN ? vown[nd][ks] : vf[nd][ks], pf[qt][ks], xa[qt][nd]);
        }
      }
      // x tile block (frame f, k-block = my head): row = query, 16-byte chunk g holds channels in perm32 order
#pragma unroll
      for (int qt = 0; qt < MT; ++qt) {
        const int row = qt * 16 + fi;
        float v[8] = {xa[qt][0][0] * inv[qt], xa[qt][0][1] * inv[qt], xa[qt][0][2] * inv[qt], xa[qt][0][3] * inv[qt],
                      xa[qt][1][0] * inv[qt], xa[qt][1][1] * inv[qt], xa[qt][1][2] * inv[qt], xa[qt][1][3] * inv[qt]};
        *reinterpret_cast<u16x8*>(xt + ((f * 8 + wave) * ROWS + row) * 32 + swz_chunk(row, fg) * 8) = cvt8<BF>(v);
      }
      lds_fence();                       // per frame: the LDS stores of several frames must never pile up (4-bit lgkmcnt)
#ifdef AXVS_STAMPS
      {   // by visit order (MQ == 2 starts with its own frame)
        const int v_ = MQ == 2 ? nvis_++ : f;
        if (v_ == 0) { AXVS_STAMP(12); } else if (v_ == 1) { AXVS_STAMP(13); } else if (v_ == 2) { AXVS_STAMP(14); } else { AXVS_STAMP(15); }
      }
#endif
    };
    {
      using I0 = std::integral_constant<int, 0>;
      using I1 = std::integral_constant<int, 1>;
      using F_ = std::false_type;
      using T_ = std::true_type;
      if constexpr (MQ == 2) {
        // my tile is frame `fown` of its sequence: visit own, own + 1, ... (mod T)
        const int fown = (int)(m0 - seq0) / L;
        auto fr = [&](int i) { const int f = fown + i; return f >= T ? f - T : f; };
        frame(fr(0), fr(1), I0{}, F_{}, T_{}, F_{});
        if constexpr (T > 2) {
          frame(fr(1), fr(2), I1{}, F_{}, F_{}, T_{});
          int i = 2;
#pragma nounroll
          for (; i + 2 <= T - 1; i += 2) {
            frame(fr(i), fr(i + 1), I0{}, F_{}, F_{}, F_{});
            frame(fr(i + 1), fr(i + 2), I1{}, F_{}, F_{}, F_{});
          }
          if constexpr (((T - 3) & 1) != 0) frame(fr(T - 2), fr(T - 1), std::integral_constant<int, (T - 2) & 1>{}, F_{}, F_{}, F_{});
          frame(fr(T - 1), 0, std::integral_constant<int, (T - 1) & 1>{}, T_{}, F_{}, F_{});
        } else {
          frame(fr(1), 0, I1{}, T_{}, F_{}, T_{});
        }
      } else {
      int f = 0;
#pragma nounroll
      for (; f + 2 <= T - 1; f += 2) {
        frame(f, f + 1, I0{}, F_{}, F_{}, F_{});
        frame(f + 1, f + 2, I1{}, F_{}, F_{}, F_{});
      }
      if constexpr (((T - 1) & 1) != 0) frame(T - 2, T - 1, I0{}, F_{}, F_{}, F_{});
      frame(T - 1, 0, std::integral_constant<int, (T - 1) & 1>{}, T_{}, F_{}, F_{});
      }
    }
    if constexpr (MQ == 0) {
      if (spatial_only) return;          // bench.py times QK^T / softmax / AV alone with this (nothing is written)
    }
  } else {
  // ---- stage the x tile: T*8 blocks of ROWS rows x 64 B, contiguous in global memory ----
  {
    constexpr int NCH = T * 8 * ROWS * 4;                       // 16-byte chunks
    constexpr int PER = (NCH + 511) / 512;                      // <= 16 for the instantiated (T, MT)
    const int mrow_max = nvalid - 1;                            // last valid row of this workgroup
    u16x8 v[PER];
#pragma unroll
    for (int p = 0; p < PER; ++p) {                             // every load in flight before the first LDS write
      const int c = min(tid + p * 512, NCH - 1);
      const int g = c & 3, row = (c >> 2) % ROWS, blk = (c >> 2) / ROWS;   // blk = f*8 + kb
      const int f = blk >> 3, kb = blk & 7;
      v[p] = *reinterpret_cast<const u16x8*>(X16 + (((long long)(kb * T + f)) * Mp + m0 + min(row, mrow_max)) * 32 + g * 8);
    }
#pragma unroll
    for (int p = 0; p < PER; ++p) {
      const int c = tid + p * 512;
      if (c < NCH) {
        const int g = c & 3, row = (c >> 2) % ROWS, blk = (c >> 2) / ROWS;
        *reinterpret_cast<u16x8*>(xt + (blk * ROWS + row) * 32 + swz_chunk(row, g) * 8) = v[p];
      }
      if ((p & 7) == 7) lds_fence();
    }
  }
  }
  if constexpr (NKS == 0) stage_small();   // (with the spatial half in the kernel this happened before its first frame)
  __syncthreads();
  if constexpr (MQ != 0) {                 // depart: every wave of this workgroup is past its poll
    const int tps = (N + ROWS - 1) / ROWS;
    if (tid == 0) atomicInc(oq.sync + m0 / N, 2 * (unsigned)tps - 1);
    if (spatial_only) return;              // measurement: q/k/v + QK^T / softmax / AV (nothing else is written)
  }

  AXVS_STAMP(1);
  // per-lane query bookkeeping
  int bown[MT], bfr[MT];                  // element offsets of my B fragment rows: own-frame slot / frame 0
  {
    // own frame of every tile row: lane l computes it for row l (two integer divisions, once), ds_bpermute hands lane
    // (fi, fg) the value of row mt*16 + fi
    const int mrow = (int)m0 + min(lane, nvalid - 1);
    const int fown_l = (int)(((unsigned)mrow % (unsigned)N) / (unsigned)L);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int row = mt * 16 + fi;
      const int fown = __builtin_amdgcn_ds_bpermute(row * 4, fown_l);
      bown[mt] = xt_off<MT>(fown, 0, row, fg);
      bfr[mt] = xt_off<MT>(0, 0, row, fg);
    }
  }
  constexpr int KBS = ROWS * 32;          // k-block stride (elements)
  constexpr int FS = 8 * ROWS * 32;       // frame stride

  // ---- q2 = scale * (Wpq_h x_own + bpq_h); refill the fragment set with Wk2_h as it goes ----
  f32x4 q2[2][MT];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < MT; ++b) q2[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  sweep8<BF, MT, true>(q2, wf, xt, bown, KBS, Wk2T, 32, wave * C, fi, fg);      // refill: slot (nt, j) <- Wk2_h^T rows of channel block j
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const float4 b = *reinterpret_cast<const float4*>(sbias + wave * 32 + nt * 16 + fg * 4);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      q2[nt][mt][0] = (q2[nt][mt][0] + b.x) * scale; q2[nt][mt][1] = (q2[nt][mt][1] + b.y) * scale;
      q2[nt][mt][2] = (q2[nt][mt][2] + b.z) * scale; q2[nt][mt][3] = (q2[nt][mt][3] + b.w) * scale;
    }
  }

  AXVS_STAMP(2);
  // ---- temporal logits and weighted values, REASSOCIATED so that proj_kv is applied once instead of once per frame:
  //        logit_f = q2 . (Wk2_h x_f) = (Wk2_h^T q2) . x_f            o = sum_f a_f (Wv2_h x_f) = Wv2_h (sum_f a_f x_f)
  //      (the reference computes k2, v2 = proj_kv(x) for all T slots, WC/temporal_attention.py:66-73: 4 T C^2 MACs per token;
  //      here 2 C^2 plus 2 T C dot / axpy work on the x tile).  Both rely on the x tile's perm32 channel order: an MFMA D tile
  //      pair (channels of block kb) and a B fragment of that block hold the same 8 channels in the same lane.
  //      The k2 bias still drops out of the softmax over f, the v2 bias is added once.
  //      x fragments are read one (channel block, row tile) group of T at a time, one group ahead (<= 2T LDS reads in flight).
  u16x8 q2f[MT];                                            // q2 as the B operand (K = head dim, perm32 order)
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    float v[8] = {q2[0][mt][0], q2[0][mt][1], q2[0][mt][2], q2[0][mt][3], q2[1][mt][0], q2[1][mt][1], q2[1][mt][2], q2[1][mt][3]};
    q2f[mt] = cvt8<BF>(v);
  }
  float lg[T][MT];
  f32x4 lacc[T][MT];                                        // MFMA tiles whose diagonals are the logits (see below)
#pragma unroll
  for (int f = 0; f < T; ++f)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) lacc[f][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    constexpr int PD = T <= 4 ? 2 : 1;                      // groups of x fragments in flight ahead of their use
    u16x8 xg[PD + 1][T];                                    // x fragments of group g live in xg[g % (PD + 1)] (no copies)
#pragma unroll
    for (int g0 = 0; g0 < PD; ++g0)
#pragma unroll
      for (int f = 0; f < T; ++f) xg[g0][f] = *reinterpret_cast<const u16x8*>(xt + bfr[g0 % MT] + f * FS + (g0 / MT) * KBS);
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
      f32x4 qk[2][MT];                                      // (Wk2_h^T q2)[channels of block kb][token]
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) qk[nt][mt] = H16<BF>::mfma(wf[nt][kb], q2f[mt], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) wf[nt][kb] = w_frag(Wpkv, 2 * C, kb, C + wave * 32 + nt * 16 + fi, fg);      // -> Wv2_h
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int g = kb * MT + mt;
        if (g + PD < 8 * MT) {
#pragma unroll
          for (int f = 0; f < T; ++f)
            xg[(g + PD) % (PD + 1)][f] = *reinterpret_cast<const u16x8*>(xt + bfr[(g + PD) % MT] + f * FS + ((g + PD) / MT) * KBS);
        }
        {   // logit_f += u . x_f over the 32 channels of block kb, for the 16 rows of tile mt: the DIAGONAL of a 16 x 16 x 32 MFMA with
            // A = u (rows = tokens: the accumulators of the qk MFMAs, converted, ARE that fragment) and B = x_f.  15 of 16 outputs are
            // unused, but the matrix pipe is mostly idle in this phase while the VALU was its bottleneck (round 4: 512 v_dot2 per wave
            // replaced by 128 MFMAs; -1.9 us on the width-pass kernel, and no register spills any more)
          float v8[8] = {qk[0][mt][0], qk[0][mt][1], qk[0][mt][2], qk[0][mt][3], qk[1][mt][0], qk[1][mt][1], qk[1][mt][2], qk[1][mt][3]};
          const u16x8 ua = cvt8<BF>(v8);
          {
#pragma unroll
          for (int f = 0; f < T; ++f) lacc[f][mt] = H16<BF>::mfma(ua, xg[g % (PD + 1)][f], lacc[f][mt]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);                  // keeps the LDS reads of later groups from being hoisted (lgkmcnt)
      }
    }
  }
  {
    // lane (fi, fg) holds D[4 fg + r][fi]: the diagonal element of row fi sits in lane (fi, fi >> 2), component fi & 3; every
    // lane of column fi fetches it from there
    const int sel = fi & 3, srcl = (fi + 16 * (fi >> 2)) * 4;
#pragma unroll
    for (int f = 0; f < T; ++f)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const f32x4 a = lacc[f][mt];
        const float d = sel == 0 ? a[0] : sel == 1 ? a[1] : sel == 2 ? a[2] : a[3];
        lg[f][mt] = __int_as_float(__builtin_amdgcn_ds_bpermute(srcl, __float_as_int(d)));
      }
  }
  AXVS_STAMP(3);
  // softmax over frames
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    float mx = lg[0][mt];
#pragma unroll
    for (int f = 1; f < T; ++f) mx = fmaxf(mx, lg[f][mt]);
    float s = 0.f;
#pragma unroll
    for (int f = 0; f < T; ++f) {
      lg[f][mt] = __expf(lg[f][mt] - mx);
      s += lg[f][mt];
    }
    const float inv = 1.f / s;
#pragma unroll
    for (int f = 0; f < T; ++f) lg[f][mt] *= inv;
  }

  // ---- o = Wv2_h (sum_f a_f x_f) + bv2_h; every slot is refilled with Wp[32w..32w+31] right after its use ----
  f32x4 o[2][MT];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const float4 b = *reinterpret_cast<const float4*>(sbias + C + wave * 32 + nt * 16 + fg * 4);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) o[nt][mt] = f32x4{b.x, b.y, b.z, b.w};
  }
  {
    constexpr int PD = T <= 4 ? 2 : 1;
    u16x8 xg[PD + 1][T];
#pragma unroll
    for (int g0 = 0; g0 < PD; ++g0)
#pragma unroll
      for (int f = 0; f < T; ++f) xg[g0][f] = *reinterpret_cast<const u16x8*>(xt + bfr[g0 % MT] + f * FS + (g0 / MT) * KBS);
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
      u16x8 xb[MT];                                         // sum_f a_f x_f, block kb, as the B operand
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int g = kb * MT + mt;
        if (g + PD < 8 * MT) {
#pragma unroll
          for (int f = 0; f < T; ++f)
            xg[(g + PD) % (PD + 1)][f] = *reinterpret_cast<const u16x8*>(xt + bfr[(g + PD) % MT] + f * FS + ((g + PD) / MT) * KBS);
        }
        u16x8 acc = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int f = 0; f < T; ++f) acc = axpy8<BF>(lg[f][mt], xg[g % (PD + 1)][f], acc);
        xb[mt] = acc;
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) o[nt][mt] = H16<BF>::mfma(wf[nt][kb], xb[mt], o[nt][mt]);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) wf[nt][kb] = w_frag(Wp, C, kb, wave * 32 + nt * 16 + fi, fg);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  AXVS_STAMP(4);
  // ---- o (all heads) -> LDS as the [8][ROWS][32] tile of the output projection (aliases the x tile) ----
  __syncthreads();                        // every wave is done reading x
  AXVS_STAMP(5);
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int n = wave * 32 + nt * 16 + fg * 4, row = mt * 16 + fi;
      *reinterpret_cast<u16x4*>(xt + ((n >> 5) * ROWS + row) * 32 + swz_chunk(row, (n & 31) >> 3) * 8 + (n & 7)) = cvt4<BF>(o[nt][mt]);
    }
  AXVS_STAMP(9);
  __syncthreads();
  AXVS_STAMP(6);
  // ---- out[:, 32w..32w+31] = Wp[32w.., :] . o + bp ----
  f32x4 po[2][MT];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < MT; ++b) po[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  int bo[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int row = mt * 16 + fi;
    bo[mt] = row * 32 + swz_chunk(row, fg) * 8;
  }
  // residual rows of the row-wise epilogue: requested before the sweep, so their HBM latency hides behind it
  constexpr int RPW = ROWS / 8;                     // rows per wave
  float4 rres[RPW];
  long long roff[RPW];
  // RowMap arithmetic (integer divisions) once per wave, lane i computing row i of the wave's RPW rows, then broadcast with
  // v_readlane -- instead of RPW unrolled copies of the same ~75-instruction sequence
  [[maybe_unused]] int roff_lo = 0, roff_hi = 0;    // element offset of row (lane % RPW) of this wave, as two words (re-read with v_readlane)
  unsigned rok;                                     // bit i: row i of this wave exists (inside its sequence, not a padding row of its frame) -> stored
  {
    const int myrow = wave * RPW + (lane % RPW);
    const int mym = (int)m0 + min(myrow, nvalid - 1);
    rok = (unsigned)__builtin_amdgcn_ballot_w64(myrow < nvalid && row_exists(rm, mym)) & ((1u << RPW) - 1u);
    const long long myoff = nat_row(rm, mym) * C;
    const int lo = (int)(myoff & 0xffffffffll), hi = (int)(myoff >> 32);
    roff_lo = lo;
    roff_hi = hi;
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const long long o = ((long long)__builtin_amdgcn_readlane(hi, i) << 32) | (unsigned)__builtin_amdgcn_readlane(lo, i);
      roff[i] = o + lane * 4;                       // wave-uniform row, lane = float4 column
      rres[i] = *reinterpret_cast<const float4*>(res + roff[i]);
    }
  }
  // with the FFN half following, the sweep leaves the first linear1 fragment set behind
  constexpr int crot = 0;                          // fixed chunk order: results do not depend on the tile index (see ffn_fused_kernel)
  if constexpr (FFN) sweep8<BF, MT, true>(po, wf, xt, bo, KBS, fa.W1, fa.F, crot * 256 + wave * 32, fi, fg);
  else sweep8<BF, MT, false>(po, wf, xt, bo, KBS, Wp, C, 0, fi, fg);
  AXVS_STAMP(7);
  // ---- row-wise epilogue: accumulators (+bias) -> LDS fp32 tile (behind the o tile) -> whole rows: + residual -> out ----
  float* etile = reinterpret_cast<float*>(xt + 8 * ROWS * 32);
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int n = wave * 32 + nt * 16 + fg * 4;
    const float4 b = *reinterpret_cast<const float4*>(sbias + 2 * C + n);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      epi_put(etile, mt * 16 + fi, n, f32x4{po[nt][mt][0] + b.x, po[nt][mt][1] + b.y, po[nt][mt][2] + b.z, po[nt][mt][3] + b.w});
  }
  __syncthreads();
  [[maybe_unused]] NoRows yrows;                   // FFN: the wave's rows after the residual
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const int row = wave * RPW + i;
    const float4 v = *reinterpret_cast<const float4*>(etile + row * kEpiLd + lane * 4);
    float4 y = float4{v.x + rres[i].x, v.y + rres[i].y, v.z + rres[i].z, v.w + rres[i].w};
    if constexpr (!FFN) {
      if (ln_g) {      // post-norm layer (cross-clip TrajectoryAttentionLayer.forward_post, CC/...:156-161): LayerNorm(x + attn(x)), eps 1e-5
        const float mu = wave_sum(y.x + y.y + y.z + y.w) * (1.f / C);
        const float d0 = y.x - mu, d1 = y.y - mu, d2 = y.z - mu, d3 = y.w - mu;
        const float rstd = rsqrtf(wave_sum(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3) * (1.f / C) + 1e-5f);
        const float4 g = *reinterpret_cast<const float4*>(ln_g + lane * 4), b = *reinterpret_cast<const float4*>(ln_b + lane * 4);
        y = float4{d0 * rstd * g.x + b.x, d1 * rstd * g.y + b.y, d2 * rstd * g.z + b.z, d3 * rstd * g.w + b.w};
      }
    }
    if constexpr (FFN) yrows.v[i] = y;             // stays on the CU: input row of the FFN half, handed to its norm1 in registers
    else if ((rok >> i) & 1u) {
      if (wt & 1) WtBuf(out).store16((unsigned)(roff[i] * 4), y);
      else *reinterpret_cast<float4*>(out + roff[i]) = y;
    }
  }
  AXVS_STAMP(8);
  if constexpr (FFN) {
    static_assert(RPW == 8, "the FFN half walks 8 rows per wave");
    auto row_off_ = [=](int row) { return ((rok >> (row % RPW)) & 1u) ? roff[row % RPW] - lane * 4 : -1ll; };       // rows of this wave
    ffn_body<BF, decltype(row_off_), false, true>(fl, wf, fa.W1, fa.W2, out, row_off_, fa.F, 0, crot, tid, wt, yrows);
  }
  AXVS_STAMP(10);
  AXVS_WG_END(FFN ? 0 : 1);
#if !defined(AXVS_STAMPS_QKV) && !defined(AXVS_STAMPS_FFN)
  if constexpr (MQ != 0 && !FFN) AXVS_STAMP_FLUSH_AT(32, 24);      // the height-pass kernel's stamps: slots 32 .. 55 (the width pass's stay in 0 .. 23)
  else if constexpr (MQ != 0) AXVS_STAMP_FLUSH(24);
  else AXVS_STAMP_FLUSH(16);
#endif
}


}  // namespace axvs

// diagnostic builds: -DAXVS_STAMPS plus -DAXVS_STAMPS_QKV stamps the QKV kernel instead of the trajectory kernel
#if defined(AXVS_STAMPS) && defined(AXVS_STAMPS_QKV)
#define QSTAMP_DECL AXVS_STAMP_DECL
#define QSTAMP(s) AXVS_STAMP(s)
#define QSTAMP_FLUSH(n) AXVS_STAMP_FLUSH(n)
#else
#define QSTAMP_DECL
#define QSTAMP(s)
#define QSTAMP_FLUSH(n)
#endif
namespace axvs {

// =====================================================================================================
// q/k/v projections of one pass, C = 256, 8 heads:  q = (Wq (src+pos) + bq) * scale*log2e,  k = Wk (src+pos) + bk,
// v = Wv src + bv  (WC/temporal_attention.py:42-44 with query = key = src+pos, value = src, :200-203).
// Workgroup = 64 sequence-order rows; the fp32 token rows are gathered through the RowMap (1 KiB contiguous each),
// converted once into two LDS tiles (src+pos | src); wave w produces head w of q, k and v (blocked [8][Mp][32] outputs).
// =====================================================================================================
template <bool BF>
__global__ __launch_bounds__(512) void qkv_fused_kernel(const float* __restrict__ src, const float* __restrict__ pos, RowMap rm,
                                                        const u16* __restrict__ Wq, const u16* __restrict__ Wk,
                                                        const u16* __restrict__ Wv, const float* __restrict__ bq,
                                                        const float* __restrict__ bk, const float* __restrict__ bv,
                                                        u16* __restrict__ Q16, u16* __restrict__ K16, u16* __restrict__ V16,
                                                        long long Mp, float qscale, u16* __restrict__ VT16, int N, int L, int T,
                                                        int NKS, PosGen pg, int wt /* write-through q/k/V^T stores (offsets < 4 GiB) */,
                                                        int* __restrict__ status /* nullable: bit 0 <- an operand left the fp16 range */) {
  constexpr int C = 256, MT = 4, ROWS = 64, KBS = ROWS * 32;
  extern __shared__ __attribute__((aligned(16))) u16 smem[];
  u16* tqk = smem;                       // (src + pos) tile [8][64][32]
  u16* tv = smem + 8 * KBS;              // src tile
  float* sbias = reinterpret_cast<float*>(smem + 16 * KBS);   // bq | bk | bv
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fi = lane & 15, fg = lane >> 4;
  const long long m0 = (long long)blockIdx.x * ROWS;

  QSTAMP_DECL;
  QSTAMP(0);
  const WtBuf wq(Q16), wk(K16), wvt(VT16), wvr(V16);
  // gridDim.y == 3 (few row tiles: the cross-clip modules' 512 clip queries): workgroup (tile, part) computes only q, k or v, so
  // each one streams a third of the weights -- with 8 tiles the chip is empty and the per-CU weight stream is the whole cost
  const bool split = gridDim.y == 3;
  const int part = split ? (int)blockIdx.y : 0;
  u16x8 wf[2][8];
  load_wfrags<2, 8>(wf, part == 0 ? Wq : part == 1 ? Wk : Wv, C, 0, wave * 32, fi, fg);
  float bias3[3];                        // biases requested now, parked in LDS behind the row gather
  if (tid < C) {
    bias3[0] = bq[tid]; bias3[1] = bk[tid]; bias3[2] = bv[tid];
  }

  // ---- gather + convert the 64 token rows: thread -> (row, float4 column), 64 consecutive threads cover one row ----
  {
    const int c4 = tid & 63;             // float4 index within the row
    // RowMap arithmetic once per wave: lane k computes the k-th of the wave's 8 rows (rows wave + 8k), v_readlane broadcasts
    int off_lo, off_hi, coords = 0;
    {
      const int myrow = (tid >> 6) + 8 * (lane & 7);
      const long long mym = min(m0 + myrow, Mp - 1);
      const long long myoff = (pg.mode ? nat_row_coords(rm, (int)mym, pg.l_is_h, &coords) : nat_row(rm, (int)mym)) * C;
      off_lo = (int)(myoff & 0xffffffffll);
      off_hi = (int)(myoff >> 32);
    }
    PosGenLane pl;
    if (pg.mode) pl.init(pg, c4 * 4);
    float amax = 0.f;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      float4 a[4], p[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = half * 4 + i;
        const long long off = (((long long)__builtin_amdgcn_readlane(off_hi, k) << 32) | (unsigned)__builtin_amdgcn_readlane(off_lo, k)) + c4 * 4;
        a[i] = *reinterpret_cast<const float4*>(src + off);
        if (pg.mode) p[i] = pl.eval(pg, __builtin_amdgcn_readlane(coords, k));      // sine embedding generated, not read
        else p[i] = pos ? *reinterpret_cast<const float4*>(pos + off) : float4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = (tid >> 6) + 8 * (half * 4 + i);
        const int n = c4 * 4, kb = n >> 5, k = n & 31;
        const int o = (kb * ROWS + row) * 32 + swz_chunk(row, k >> 3) * 8 + (k & 7);
        *reinterpret_cast<u16x4*>(tv + o) = cvt4<BF>(f32x4{a[i].x, a[i].y, a[i].z, a[i].w});
        *reinterpret_cast<u16x4*>(tqk + o) = cvt4<BF>(f32x4{a[i].x + p[i].x, a[i].y + p[i].y, a[i].z + p[i].z, a[i].w + p[i].w});
        if (!BF) {     // fp16 operands: remember the largest magnitude that gets rounded (range check below)
          amax = fmaximum(amax, fmaximum(fmaximum(fabsf(a[i].x), fabsf(a[i].y)), fmaximum(fabsf(a[i].z), fabsf(a[i].w))));
          amax = fmaximum(amax, fmaximum(fmaximum(fabsf(a[i].x + p[i].x), fabsf(a[i].y + p[i].y)), fmaximum(fabsf(a[i].z + p[i].z), fabsf(a[i].w + p[i].w))));
        }
      }
      lds_fence();
    }
    // a value beyond the fp16 range (or a NaN) would silently become inf in the MFMA operands: report it (axvs_set_status_buffer)
    if (!BF && status != nullptr && !(amax <= 65504.f)) atomicOr(status, 1);
  }
  if (tid < C) {
    sbias[tid] = bias3[0];
    sbias[C + tid] = bias3[1];
    sbias[2 * C + tid] = bias3[2];
  }
  QSTAMP(1);
  __syncthreads();
  QSTAMP(2);

  int bb[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int row = mt * 16 + fi;
    bb[mt] = row * 32 + swz_chunk(row, fg) * 8;
  }
  // q, k from the (src+pos) tile, v from the src tile; each sweep refills the fragment set for the next one
#pragma unroll
  for (int which = 0; which < 3; ++which) {
    if (split && which != part) continue;
    f32x4 acc[2][MT];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < MT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (split && which < 2) sweep8<BF, MT, false>(acc, wf, tqk, bb, KBS, Wk, C, 0, fi, fg);        // no refill: nothing follows
    else if (which == 0) sweep8<BF, MT, true>(acc, wf, tqk, bb, KBS, Wk, C, wave * 32, fi, fg);
    else if (which == 1) sweep8<BF, MT, true>(acc, wf, tqk, bb, KBS, Wv, C, wave * 32, fi, fg);
    else if (VT16 == nullptr) sweep8<BF, MT, false>(acc, wf, tv, bb, KBS, Wv, C, 0, fi, fg);
    else sweep8<BF, MT, false, true>(acc, wf, tv, bb, KBS, Wv, C, 0, fi, fg);          // tokens on D rows
    if (which == 0) QSTAMP(3);
    if (which == 1) QSTAMP(5);
    if (which == 2) QSTAMP(7);
    if (which == 2 && VT16 != nullptr) {
      // block-transposed V^T:  VT[head][frame slot sf = m'/L][ks][nd][16 d][32 keys in perm32 order]; a lane holds channel
      // nt*16+fi and tokens 4g..4g+3 of tile mt (one frame: L % 16 == 0) -> 8 contiguous bytes
      const long long heads_sf = (unsigned)Mp / (unsigned)L;  // S*T frame slots
      if (L % 32 == 0) {
        // tile pairs (mt, mt+1) are the two 16-key halves of one 32-key step: 16 contiguous bytes per lane
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const float b = sbias[2 * C + wave * 32 + nt * 16 + fi];
#pragma unroll
          for (int mp = 0; mp < MT; mp += 2) {
            const unsigned mt0 = (unsigned)m0 + mp * 16;         // row indices fit 32 bits (checked by the host)
            if (mt0 < (unsigned)Mp) {
              const unsigned sf = mt0 / (unsigned)L;
              const int ks = (int)(mt0 - sf * L) >> 5;
              const long long d = ((((long long)wave * heads_sf + sf) * NKS + ks) * 2 + nt) * 512 + fi * 32 + fg * 8;
              float v[8] = {acc[nt][mp][0] + b, acc[nt][mp][1] + b, acc[nt][mp][2] + b, acc[nt][mp][3] + b,
                            acc[nt][mp + 1][0] + b, acc[nt][mp + 1][1] + b, acc[nt][mp + 1][2] + b, acc[nt][mp + 1][3] + b};
              if (wt) wvt.store16((unsigned)(d * 2), cvt8<BF>(v));
              else *reinterpret_cast<u16x8*>(VT16 + d) = cvt8<BF>(v);
            }
          }
        }
      } else if (L % 16 != 0) {
        // any frame length: the 4 tokens of a lane may sit in different frames (even sequences) -> one 2-byte store per token.
        // Key l of a frame goes to 32-key step l >> 5, position ((kk >> 2) & 3) * 8 + (kk >> 4) * 4 + (kk & 3), kk = l & 31 (the
        // k-permuted order of the P^T operand, axvs_attn.h).
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          unsigned tok = (unsigned)m0 + mt * 16 + fg * 4;
          unsigned sf = tok / (unsigned)L;
          int l = (int)(tok - sf * L);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (tok < (unsigned)Mp) {
              const int ks = l >> 5, kk = l & 31;
              const int pp = ((kk >> 2) & 3) * 8 + (kk >> 4) * 4 + (kk & 3);
#pragma unroll
              for (int nt = 0; nt < 2; ++nt) {
                const long long d = ((((long long)wave * heads_sf + sf) * NKS + ks) * 2 + nt) * 512 + fi * 32 + pp;
                VT16[d] = H16<BF>::from_f32(acc[nt][mt][r] + sbias[2 * C + wave * 32 + nt * 16 + fi]);
              }
              // the frame's last key also clears the padding keys L .. 32 NKS - 1 of its d rows (they get probability 0 but must be
              // finite): no memset of the V^T buffer per pass
              if (l == L - 1) {
                for (int l2 = L; l2 < NKS * 32; ++l2) {
                  const int k2 = l2 & 31, p2 = ((k2 >> 2) & 3) * 8 + (k2 >> 4) * 4 + (k2 & 3);
#pragma unroll
                  for (int nt = 0; nt < 2; ++nt)
                    VT16[((((long long)wave * heads_sf + sf) * NKS + (l2 >> 5)) * 2 + nt) * 512 + fi * 32 + p2] = 0;
                }
              }
            }
            ++tok;
            if (++l == L) { l = 0; ++sf; }
          }
        }
      } else {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const float b = sbias[2 * C + wave * 32 + nt * 16 + fi];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const unsigned mt0 = (unsigned)m0 + mt * 16;        // first token of the tile
          if (mt0 < (unsigned)Mp) {
            const unsigned sf = mt0 / (unsigned)L;
            const int l = (int)(mt0 - sf * L) + fg * 4;       // key index within the frame of the lane's first token
            const int ks = l >> 5, pp = fg * 8 + ((l >> 4) & 1) * 4;
            const long long d = ((((long long)wave * heads_sf + sf) * NKS + ks) * 2 + nt) * 512 + fi * 32 + pp;
            f32x4 v = acc[nt][mt];
            v[0] += b; v[1] += b; v[2] += b; v[3] += b;
            if (wt) wvt.store8((unsigned)(d * 2), cvt4<BF>(v));
            else *reinterpret_cast<u16x4*>(VT16 + d) = cvt4<BF>(v);
            // frame length an odd multiple of 16: the frame's last 16-key tile also clears the other (padding) half of its
            // 32-key step -- pad keys get probability 0 but must hold finite values (this replaces a memset per pass)
            if (L % 32 != 0 && (int)(mt0 - sf * L) + 16 == L) *reinterpret_cast<u16x4*>(VT16 + (d ^ 4)) = u16x4{0, 0, 0, 0};
          }
        }
      }
      }
      QSTAMP(8);
      continue;
    }
    u16* dst = which == 0 ? Q16 : which == 1 ? K16 : V16;
    const float sc = which == 0 ? qscale : 1.f;
    const float4 b0 = *reinterpret_cast<const float4*>(sbias + which * C + wave * 32 + fg * 4);
    const float4 b1 = *reinterpret_cast<const float4*>(sbias + which * C + wave * 32 + 16 + fg * 4);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const long long m = m0 + mt * 16 + fi;
      if (m < Mp) {
        if (which < 2) {   // q, k: stored position g*8 + nt*4 + r  <-  channel nt*16 + 4g + r: 16 contiguous bytes per lane
          float v[8] = {(acc[0][mt][0] + b0.x) * sc, (acc[0][mt][1] + b0.y) * sc, (acc[0][mt][2] + b0.z) * sc,
                        (acc[0][mt][3] + b0.w) * sc, (acc[1][mt][0] + b1.x) * sc, (acc[1][mt][1] + b1.y) * sc,
                        (acc[1][mt][2] + b1.z) * sc, (acc[1][mt][3] + b1.w) * sc};
          const long long d = ((long long)wave * Mp + m) * 32 + fg * 8;
          if (wt) (which == 0 ? wq : which == 1 ? wk : wvr).store16((unsigned)(d * 2), cvt8<BF>(v));
          else *reinterpret_cast<u16x8*>(dst + d) = cvt8<BF>(v);
        } else {           // v: natural channel order
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            const float4 b = *reinterpret_cast<const float4*>(sbias + 2 * C + wave * 32 + nt * 16 + fg * 4);
            f32x4 v = acc[nt][mt];
            v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
            *reinterpret_cast<u16x4*>(dst + ((long long)wave * Mp + m) * 32 + nt * 16 + fg * 4) = cvt4<BF>(v);
          }
        }
      }
    }
    if (which == 0) QSTAMP(4);
    if (which == 1) QSTAMP(6);
  }
  QSTAMP_FLUSH(9);
}

constexpr size_t kQkvLdsBytes = 16 * 64 * 32 * sizeof(u16) + 3 * 256 * sizeof(float);

}  // namespace axvs
