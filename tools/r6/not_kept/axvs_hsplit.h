// Head-split trajectory attention for FEW rows (round 6): one launch per pass, workgroup = (16-row tile, head).
//
// WC/temporal_attention.py:35-76 (TrajectoryAttention.forward) + the residual (:200-213) [+ the cross-clip layer's post-norm,
// CC/maxtron_cross_clip_tracking_module.py:156-161], C = 256, 8 heads of 32 channels, 16-bit MFMA operands, fp32 accumulation.
//
// Why.  With a few hundred to a few thousand rows (the cross-clip queries: 512 rows per video; the coarse pyramid levels of the shipped
// configurations: 25 x 43 x 2, 12 x 20 x 5) temporal_fused_kernel (axvs_fused.h) gives every 16-row tile ONE workgroup whose 8 waves each own a
// head: that workgroup streams the whole weight set of the pass (q/k/v 384 KiB + proj_q / proj_kv / proj 512 KiB) through ONE CU's L2 port
// (~60 B/clk: 15 k cycles), on 32 .. 170 of the 256 CUs (profiles/r5_few_rows_timeline.txt).  Here the 8 heads of a tile are 8 workgroups on 8
// CUs, each streaming its head's 112 KiB, and a tile's frames run on the 4 waves of a workgroup side by side instead of one after the other.
// What the heads of a tile share travels through memory inside the launch, write-through (sc1) stores and sc1 loads (MI355X_MICROARCH.md,
// inter-workgroup visibility: every storing wave drains vmcnt, workgroup barrier, ONE lane adds to the counter; consumers poll it with sc1 loads):
//   1. K / V^T of the (sequence, head): written by the row tiles of the sequence, read by all of them        [counter per (sequence, head)]
//   2. x[row, frame, 32 channels of the head] -> the tile's T x 256 x-rows, read by the 8 heads of the tile      [counter per tile]
//   3. the head's K = 32 slice of the output projection, fp32 [8][rows][256] -> added IN HEAD ORDER by the workgroup that owns the row
//      (2 rows of the tile per head), + bias + residual [+ LayerNorm]                                             [counter per tile]
// blockIdx = tile * 8 + head: the workgroups of a head share an XCD under round-robin placement (its weights and hand-off 1 stay in one L2;
// speed only).  Every counter receives `n` arrivals and `n` departures through atomicInc(.., 2 n - 1): zero before the launch, zero after it.
// A wait that runs out sets status bit 2 and POISONS what the workgroup hands on (NaN), so the rows concerned come out NaN -- never numbers
// computed from stale bytes; nobody waits for a workgroup that has not been dispatched behind it in its own sequence (deadlock-free for in-order
// dispatch as long as one sequence's workgroups fit the chip: the host checks).
//
// Row space: the PADDED sequence order of the fused tier (RowMap: frames of Lp = roundup16(L) rows, rows l >= L clamped copies that are computed
// and never stored), so a 16-row tile lies inside ONE frame of one sequence.  Results do not depend on the grid: bit-identical across batch sharding.
#pragma once
#include "axvs_common.h"
#include "axvs_fused.h"

namespace axvs {

struct HsArgs {
  const float* src;        // fp32 token rows [.., 256] addressed through rm (query = key = src + pos, value = src)
  const float* res;        // residual rows (same addressing)
  const float* pos;        // nullable; read when pg.mode == 0
  PosGen pg;
  RowMap rm;
  const u16 *Wq, *Wk, *Wv, *Wpq, *Wk2T, *Wpkv, *Wp;      // TrajPacked (axvs_host.h)
  const float *bq, *bk, *bv, *bpq, *bpkv, *bp;
  const float *ln_g, *ln_b;                               // nullable: out = LayerNorm(res + attn) (eps 1e-5)
  float* out;              // fp32 rows through rm
  u16 *Q16, *K16, *VT16;   // [8][Mp][32] perm32 (Q16: phase-by-phase launches only) | [8][frame slots][NKS][2][16][32]
  u16* X16;                // [Mp][T][256]: x rows, 32 channels of a head in perm32 order
  float* part;             // [8][Mp][256] fp32: per-head slices of the output projection
  unsigned* sync;          // [S * 8] K/V^T counters | [tiles] x counters | [tiles] projection counters
  int* status;
  unsigned spin_limit;
  long long Mp;            // padded rows (S * T * Lp)
  int S, L /* = Lp */;
  float scale, qscale;     // head_dim^-0.5 | * log2(e)
  // Phases this launch runs (bit i = phase i + 1).  15: the whole pass in ONE launch with the three in-launch hand-offs (needs `sync`).  1, 2, 4, 8 as four
  // launches: no counter, no wait -- the hand-offs are the kernel boundaries -- and q travels through Q16 instead of LDS.  Same code on the same 16-bit values:
  // the two forms are bit-identical (callers without arrival counters, graph captures that registered none, the 'verify' policy's re-run).
  int phases;
};

// T: frames per sequence; NKS: 32-key steps per frame (Lp <= 32 NKS)
template <bool BF, int T, int NKS>
__global__ __launch_bounds__(256) void hs_traj_kernel(HsArgs a) {
  constexpr int C = 256;
  __shared__ __attribute__((aligned(16))) u16 tqk[8 * 16 * 32];      // (src + pos) tile [kb][row][32], chunk-swizzled
  __shared__ __attribute__((aligned(16))) u16 tv[8 * 16 * 32];       // src tile
  __shared__ __attribute__((aligned(16))) u16 sq[16 * 32];           // q of the tile (B fragments of QK^T), later o (natural order) for the projection slice
  __shared__ float sred[4][T > 8 ? T : 8][16];                         // cross-wave reductions: temporal logits [wave][f][row]
  __shared__ __attribute__((aligned(16))) float sored[4][2][64][4];   // cross-wave reduction of o [wave][nt][lane]
  __shared__ float sb[3 * 32 + 2 * 32];                               // bq_h | bk_h | bv_h | bpq_h | bv2_h
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fi = lane & 15, fg = lane >> 4;
  const int h = blockIdx.x & 7;
  const long long tile = blockIdx.x >> 3;
  const int L = a.L, N = T * L, tps = N / 16;                          // tiles per sequence (N is a multiple of 16)
  const long long sq_idx = tile / tps;                                 // sequence
  const int n0 = (int)(tile - sq_idx * tps) * 16;                      // first row of the tile inside its sequence
  const long long m0 = sq_idx * N + n0, seq0 = sq_idx * N;
  const int fown = n0 / L, l0 = n0 - fown * L;                         // the tile's frame, its first key index in that frame
  const int Lr = a.rm.Lv ? a.rm.Lv : L;                                // keys / rows of a frame that exist
  const long long Mp = a.Mp, nsf = Mp / L;                             // frame slots
  const bool one = a.phases == 15;                                      // one launch: in-launch hand-offs
  unsigned* const kvcnt = a.sync + sq_idx * 8 + h;
  unsigned* const xcnt = a.sync + (long long)a.S * 8 + tile;
  unsigned* const pcnt = a.sync + (long long)a.S * 8 + (long long)a.S * tps + tile;
  float poison = 1.f;
  auto wait_for = [&](unsigned* cnt, unsigned n) {
    unsigned spins = 0;
    while (ld_sc1_u32(cnt) < n) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > a.spin_limit) {
        if (a.status != nullptr && lane == 0) atomicOr(a.status, 4);
        poison = __uint_as_float(__builtin_amdgcn_readfirstlane(0x7fc00000u));
        break;
      }
    }
  };

  if (tid < 32) { sb[96 + tid] = a.bpq[h * 32 + tid]; sb[128 + tid] = a.bpkv[C + h * 32 + tid]; }
  __syncthreads();
  // ---------------- phase 1: q, k, v of my 16 rows, head h (WC/temporal_attention.py:42-44) ----------------
  if (a.phases & 1) {
  {
    // thread -> (row = tid >> 4, float4 columns (tid & 15) + 16 i): 256 contiguous bytes per 16 lanes
    const int row = tid >> 4, c4 = tid & 15;
    int coords = 0;
    const int mym = (int)m0 + row;
    const long long roff = (a.pg.mode ? nat_row_coords(a.rm, mym, a.pg.l_is_h, &coords) : nat_row(a.rm, mym)) * C;
    float4 x[4], p[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = *reinterpret_cast<const float4*>(a.src + roff + (c4 + 16 * i) * 4);
    if (a.pg.mode) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        PosGenLane pl;
        pl.init(a.pg, (c4 + 16 * i) * 4);
        p[i] = pl.eval(a.pg, coords);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) p[i] = a.pos ? *reinterpret_cast<const float4*>(a.pos + roff + (c4 + 16 * i) * 4) : float4{0.f, 0.f, 0.f, 0.f};
    }
    if (tid < 32) { sb[tid] = a.bq[h * 32 + tid]; sb[32 + tid] = a.bk[h * 32 + tid]; sb[64 + tid] = a.bv[h * 32 + tid]; }
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = (c4 + 16 * i) * 4, kb = n >> 5, k = n & 31;
      const int o = (kb * 16 + row) * 32 + swz_chunk(row, k >> 3) * 8 + (k & 7);
      const f32x4 xv = {x[i].x, x[i].y, x[i].z, x[i].w};
      const f32x4 xp = {x[i].x + p[i].x, x[i].y + p[i].y, x[i].z + p[i].z, x[i].w + p[i].w};
      *reinterpret_cast<u16x4*>(tv + o) = cvt4<BF>(xv);
      *reinterpret_cast<u16x4*>(tqk + o) = cvt4<BF>(xp);
      if (!BF) {
#pragma unroll
        for (int j = 0; j < 4; ++j) amax = fmaximum(amax, fmaximum(fabsf(xv[j]), fabsf(xp[j])));
      }
    }
    if (!BF && a.status != nullptr && !(amax <= 65504.f)) atomicOr(a.status, 1);
  }
  __syncthreads();
  if (wave < 3) {
    // wave 0: v (operands swapped: tokens on the D rows -> block-transposed V^T), wave 1: k, wave 2: q
    const u16* W = wave == 0 ? a.Wv : wave == 1 ? a.Wk : a.Wq;
    const u16* tile_in = wave == 0 ? tv : tqk;
    u16x8 wf[2][8], xf[8];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int kb = 0; kb < 8; ++kb) wf[nt][kb] = w_frag(W, C, kb, h * 32 + nt * 16 + fi, fg);
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) xf[kb] = *reinterpret_cast<const u16x8*>(tile_in + (kb * 16 + fi) * 32 + swz_chunk(fi, fg) * 8);
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    if (wave == 0) {
#pragma unroll
      for (int kb = 0; kb < 8; ++kb)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[nt] = H16<BF>::mfma(xf[kb], wf[nt][kb], acc[nt]);      // D[token 4 fg + r][channel nt * 16 + fi]
      const WtBuf vb(a.VT16);
      const long long sf = (seq0 + (long long)fown * L) / L;                 // frame slot
      const int ks = l0 >> 5, pp = fg * 8 + ((l0 >> 4) & 1) * 4;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const float b = sb[64 + nt * 16 + fi];
        const long long d = ((((long long)h * nsf + sf) * NKS + ks) * 2 + nt) * 512 + fi * 32 + pp;
        f32x4 v = acc[nt];
        v[0] += b; v[1] += b; v[2] += b; v[3] += b;
        vb.store8((unsigned)(d * 2), cvt4<BF>(v));
        // the frame's last 16-key tile also clears the padding half of its 32-key step (finite values for probability 0)
        if (L % 32 != 0 && l0 + 16 == L) vb.store8((unsigned)((d ^ 4) * 2), u16x4{0, 0, 0, 0});
      }
    } else {
#pragma unroll
      for (int kb = 0; kb < 8; ++kb)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[nt] = H16<BF>::mfma(wf[nt][kb], xf[kb], acc[nt]);      // D[channel nt * 16 + 4 fg + r][token fi]
      const float* bb = sb + (wave == 1 ? 32 : 0);
      const float sc_ = wave == 1 ? 1.f : a.qscale;
      float v[8];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[nt * 4 + r] = (acc[nt][r] + bb[nt * 16 + fg * 4 + r]) * sc_;
      const u16x8 o8 = cvt8<BF>(v);                                          // perm32 positions fg * 8 .. + 7 of row fi
      if (wave == 1) {
        const WtBuf kbuf(a.K16);
        kbuf.store16((unsigned)((((long long)h * Mp + m0 + fi) * 32 + fg * 8) * 2), o8);
      } else if (one) {
        *reinterpret_cast<u16x8*>(sq + fi * 32 + fg * 8) = o8;
      } else {
        *reinterpret_cast<u16x8*>(a.Q16 + (((long long)h * Mp + m0 + fi) * 32 + fg * 8)) = o8;
      }
    }
  }
  if (!one) return;
  vm_drain();
  __syncthreads();
  if (tid == 0) atomicInc(kvcnt, 2 * (unsigned)tps - 1);
  }

  // ---------------- phase 2: QK^T -> per-frame softmax -> AV, head h, wave = frame (WC/temporal_attention.py:46-57) ----------------
  if (a.phases & 2) {
  {
    const u16x8 qf = one ? *reinterpret_cast<const u16x8*>(sq + fi * 32 + fg * 8)
                         : *reinterpret_cast<const u16x8*>(a.Q16 + (((long long)h * Mp + m0 + fi) * 32 + fg * 8));
    if (one && tps > 1) wait_for(kvcnt, (unsigned)tps);
    const ScBuf kbuf(a.K16), vbuf(a.VT16);
    const WtBuf xb(a.X16);
    u16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = H16<BF>::from_f32(1.f);
    for (int f = wave; f < T; f += 4) {
      u16x8 kf[2 * NKS], vf[2][NKS];
#pragma unroll
      for (int kt = 0; kt < 2 * NKS; ++kt)
        kf[kt] = kbuf.load16((unsigned)((((long long)h * Mp + seq0 + (long long)f * L + min(kt * 16 + fi, Lr - 1)) * 32 + fg * 8) * 2), 0);
      const long long sf = seq0 / L + f;
#pragma unroll
      for (int nd = 0; nd < 2; ++nd)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
          vf[nd][ks] = vbuf.load16((unsigned)((((((long long)h * nsf + sf) * NKS + ks) * 2 + nd) * 512 + fi * 32 + fg * 8) * 2), 0);
      f32x4 sc[2 * NKS];
#pragma unroll
      for (int kt = 0; kt < 2 * NKS; ++kt) sc[kt] = H16<BF>::mfma(kf[kt], qf, f32x4{0.f, 0.f, 0.f, 0.f});      // D[key][query]
#pragma unroll
      for (int kt = 0; kt < 2 * NKS; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (kt * 16 + fg * 4 + r >= Lr) sc[kt][r] = -INFINITY;
      float mx = sc[0][0];
#pragma unroll
      for (int kt = 0; kt < 2 * NKS; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaximum(mx, sc[kt][r]);
      mx = groups_maximum(mx);
#pragma unroll
      for (int kt = 0; kt < 2 * NKS; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) sc[kt][r] = __builtin_amdgcn_exp2f(sc[kt][r] - mx);
      u16x8 pf[NKS];
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[ks][j] = H16<BF>::from_f32(sc[2 * ks + (j >> 2)][j & 3]);
      f32x4 ssum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) ssum = H16<BF>::mfma(ones, pf[ks], ssum);
      const float inv = __builtin_amdgcn_rcpf(ssum[0]) * poison;
      float v[8];
#pragma unroll
      for (int nd = 0; nd < 2; ++nd) {
        f32x4 xa = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) xa = H16<BF>::mfma(vf[nd][ks], pf[ks], xa);             // D[channel nd * 16 + 4 fg + r][query fi]
#pragma unroll
        for (int r = 0; r < 4; ++r) v[nd * 4 + r] = xa[r] * inv;
      }
      // x row (m0 + fi), frame f, head block h: perm32 positions fg * 8 .. + 7
      xb.store16((unsigned)(((((long long)m0 + fi) * T + f) * C + h * 32 + fg * 8) * 2), cvt8<BF>(v));
    }
  }
  if (!one) return;
  vm_drain();
  __syncthreads();
  if (tid == 0) {
    atomicInc(kvcnt, 2 * (unsigned)tps - 1);        // depart
    atomicInc(xcnt, 15);
  }
  }

  // ---------------- phase 3: temporal half of head h, reassociated (WC/temporal_attention.py:60-75; axvs_fused.h) ----------------
  if (a.phases & 4) {
  if (one) wait_for(xcnt, 8);
  {
    const ScBuf xs(a.X16);
    // q2 = scale * (Wpq_h x_own + bpq_h): every wave computes it for itself (16 MFMAs; no exchange)
    u16x8 q2f;
    {
      u16x8 wq[2][8], xo[8];
#pragma unroll
      for (int kb = 0; kb < 8; ++kb) xo[kb] = xs.load16((unsigned)(((((long long)m0 + fi) * T + fown) * C + kb * 32 + fg * 8) * 2), 0);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) wq[nt][kb] = w_frag(a.Wpq, C, kb, h * 32 + nt * 16 + fi, fg);
      f32x4 q2[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int kb = 0; kb < 8; ++kb)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) q2[nt] = H16<BF>::mfma(wq[nt][kb], xo[kb], q2[nt]);
      float v[8];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[nt * 4 + r] = (q2[nt][r] + sb[96 + nt * 16 + fg * 4 + r]) * a.scale;
      q2f = cvt8<BF>(v);
    }
    // my two channel blocks kb = 2 wave, 2 wave + 1: u = Wk2_h^T q2 on their 64 channels, logit_f partial = u . x_f, then z = sum_f a_f x_f
    u16x8 xfr[2][T];
    float lgp[T];
#pragma unroll
    for (int f = 0; f < T; ++f) lgp[f] = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int kb = 2 * wave + j;
#pragma unroll
      for (int f = 0; f < T; ++f) xfr[j][f] = xs.load16((unsigned)(((((long long)m0 + fi) * T + f) * C + kb * 32 + fg * 8) * 2), 0);
      f32x4 qk[2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
        qk[nt] = H16<BF>::mfma(w_frag(a.Wk2T, 32, kb, h * C + nt * 16 + fi, fg), q2f, f32x4{0.f, 0.f, 0.f, 0.f});
      float v8[8] = {qk[0][0], qk[0][1], qk[0][2], qk[0][3], qk[1][0], qk[1][1], qk[1][2], qk[1][3]};
      const u16x8 ua = cvt8<BF>(v8);
#pragma unroll
      for (int f = 0; f < T; ++f) lgp[f] = dot8_acc<BF>(ua, xfr[j][f], lgp[f]);
    }
#pragma unroll
    for (int f = 0; f < T; ++f) {
      const float s = groups_sum(lgp[f]);
      if (fg == 0) sred[wave][f][fi] = s;
    }
    __syncthreads();
    float at[T];
    {
      float mx = -INFINITY;
#pragma unroll
      for (int f = 0; f < T; ++f) {
        at[f] = (sred[0][f][fi] + sred[1][f][fi]) + (sred[2][f][fi] + sred[3][f][fi]);
        mx = fmaxf(mx, at[f]);
      }
      float s = 0.f;
#pragma unroll
      for (int f = 0; f < T; ++f) {
        at[f] = __expf(at[f] - mx);
        s += at[f];
      }
      const float is = 1.f / s;
#pragma unroll
      for (int f = 0; f < T; ++f) at[f] *= is;
    }
    // o partial over my channel blocks: Wv2_h z
    f32x4 oacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int kb = 2 * wave + j;
      u16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int f = 0; f < T; ++f) z = axpy8<BF>(at[f], xfr[j][f], z);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) oacc[nt] = H16<BF>::mfma(w_frag(a.Wpkv, 2 * C, kb, C + h * 32 + nt * 16 + fi, fg), z, oacc[nt]);
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) *reinterpret_cast<f32x4*>(&sored[wave][nt][lane][0]) = oacc[nt];
    __syncthreads();
    if (wave == 0) {
      // o = sum over the waves' channel blocks (fixed order) + bv2 -> 16-bit [row][32 channels, natural order] (sq is free: q was read)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        f32x4 o = (*reinterpret_cast<const f32x4*>(&sored[0][nt][lane][0]) + *reinterpret_cast<const f32x4*>(&sored[1][nt][lane][0])) +
                  (*reinterpret_cast<const f32x4*>(&sored[2][nt][lane][0]) + *reinterpret_cast<const f32x4*>(&sored[3][nt][lane][0]));
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (o[r] + sb[128 + nt * 16 + fg * 4 + r]) * poison;
        *reinterpret_cast<u16x4*>(sq + fi * 32 + nt * 16 + fg * 4) = cvt4<BF>(o);
      }
    }
    __syncthreads();
    // my slice of the output projection: part[h][row][c] = sum_n Wp[c][h * 32 + n] o[row][n], 16 channel tiles over 4 waves
    {
      const u16x8 of = *reinterpret_cast<const u16x8*>(sq + fi * 32 + fg * 8);
      const WtBuf pb(a.part);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ct = wave * 4 + j;
        const f32x4 d = H16<BF>::mfma(w_frag(a.Wp, C, h, ct * 16 + fi, fg), of, f32x4{0.f, 0.f, 0.f, 0.f});      // D[channel ct * 16 + 4 fg + r][row fi]
        pb.store16((unsigned)(((((long long)h * Mp + m0 + fi) * C) + ct * 16 + fg * 4) * 4), float4{d[0], d[1], d[2], d[3]});
      }
    }
  }
  if (!one) return;
  vm_drain();
  __syncthreads();
  if (tid == 0) {
    atomicInc(xcnt, 15);          // depart
    atomicInc(pcnt, 15);
  }
  }

  // ---------------- phase 4: rows 2 h, 2 h + 1 of the tile: sum of the 8 heads' slices (head order) + bias + residual [+ LayerNorm] ----------------
  if (!(a.phases & 8)) return;
  if (wave < 2) {
    if (one) wait_for(pcnt, 8);
    const int row = 2 * h + wave;
    const int mym = (int)m0 + row;
    const ScBuf ps(a.part);
    float4 pv[8];
#pragma unroll
    for (int hh = 0; hh < 8; ++hh)
      pv[hh] = __builtin_bit_cast(float4, ps.load16((unsigned)((((long long)hh * Mp + mym) * C + lane * 4) * 4), 0));
    if (row_exists(a.rm, mym)) {
      const long long off = nat_row(a.rm, mym) * C + lane * 4;
      const float4 rs = *reinterpret_cast<const float4*>(a.res + off);
      const float4 bp = *reinterpret_cast<const float4*>(a.bp + lane * 4);
      float4 acc = pv[0];
#pragma unroll
      for (int hh = 1; hh < 8; ++hh) { acc.x += pv[hh].x; acc.y += pv[hh].y; acc.z += pv[hh].z; acc.w += pv[hh].w; }
      float4 o = {(acc.x + bp.x + rs.x) * poison, (acc.y + bp.y + rs.y) * poison, (acc.z + bp.z + rs.z) * poison, (acc.w + bp.w + rs.w) * poison};
      if (a.ln_g != nullptr) {
        const float mu = wave_sum(o.x + o.y + o.z + o.w) * (1.f / C);
        const float d0 = o.x - mu, d1 = o.y - mu, d2 = o.z - mu, d3 = o.w - mu;
        const float rstd = rsqrtf(wave_sum(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3) * (1.f / C) + 1e-5f);
        const float4 g = *reinterpret_cast<const float4*>(a.ln_g + lane * 4), b = *reinterpret_cast<const float4*>(a.ln_b + lane * 4);
        o = float4{d0 * rstd * g.x + b.x, d1 * rstd * g.y + b.y, d2 * rstd * g.z + b.z, d3 * rstd * g.w + b.w};
      }
      *reinterpret_cast<float4*>(a.out + off) = o;
    }
  }
  if (!one) return;
  __syncthreads();
  if (tid == 0) atomicInc(pcnt, 15);          // depart: the counter is zero again after the 16th add
}

}  // namespace axvs
