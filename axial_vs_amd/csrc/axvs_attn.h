// Spatial half of trajectory attention (WC/temporal_attention.py:46-57):
//   for every query q of a sequence and every frame f:  x[q, f, :] = softmax_l(scale * q . k[f, l]) @ v[f, l, :]
// One softmax per (query, frame) -- the N x N logits never leave registers.
//
// Output x: 16-bit [heads][T][Mtot][32] (frame-major planes of the blocked layout), the 32 channels of a head block in
// perm32 order (axvs_misc.h) so every lane stores 16 contiguous bytes.
// Inputs are the blocked 16-bit q/k/v written by the QKV GEMM ([heads][Mtot][32], sequence-order rows, q already
// multiplied by scale*log2(e)).  Workgroup = (up to 256 queries, head, sequence); wave = 32 queries.  K and V of the
// (sequence, head) live in LDS for the whole workgroup.
//
// MFMA orientation ("keys on rows"):  S^T = K . Q^T  gives D[key][query] with the query on the lane (lane&15) and
// the keys in registers/lane-groups, so the softmax statistics are in-lane reductions plus two cross-lane steps, and
// the exponentiated tile is directly the B operand of  x^T = V^T . P^T  (accumulator-as-operand, k-permuted: slot
// (g, j) of a 32-key step holds key 16*(j>>2) + 4*g + (j&3); V^T fragments follow the same permutation through
// ds_read_b64_tr_b16).  x^T = D[d][query] leaves 4 consecutive channels of one (query, frame) row per lane.
#pragma once
#include "axvs_common.h"

namespace axvs {

// LDS images (16-bit elements):  K: [T*LP][32] rows with swz_chunk;  V: [T*LP][32] rows, the two 16-column halves
// swapped on rows with (key>>2)&1 so the transposed reads of a 32-lane half hit distinct banks.
__device__ __forceinline__ int v_lds_off(int key, int dcol) {
  return key * 32 + (((dcol >> 4) ^ ((key >> 2) & 1)) << 4) + (dcol & 15);
}

template <bool BF, int NKS>  // NKS = 32-key steps per frame; LP = 32*NKS >= L
__global__ __launch_bounds__(512) void spatial_attn_kernel(const u16* __restrict__ Q16, const u16* __restrict__ K16,
                                                           const u16* __restrict__ V16, u16* __restrict__ X,
                                                           float* __restrict__ attn, int N, int T, int L, int heads,
                                                           long long Mtot, int TCH /* frames whose K / V fit the LDS at a time */) {
  extern __shared__ __attribute__((aligned(16))) u16 smem[];
  constexpr int LP = NKS * 32;
  constexpr int QT = 2;                         // 16-query tiles per wave: every K / V fragment read feeds two MFMAs
  u16* sK = smem;
  u16* sV = smem + (size_t)TCH * LP * 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthreads = blockDim.x;
  const int h = blockIdx.y, s = blockIdx.z;
  const int fi = lane & 15, fg = lane >> 4;
  const long long seq0 = (long long)s * N;
  const u16* Kh = K16 + (long long)h * Mtot * 32;
  const u16* Vh = V16 + (long long)h * Mtot * 32;
  const u16* Qh = Q16 + (long long)h * Mtot * 32;

  const int q0 = (blockIdx.x * (nthreads >> 6) + wave) * (16 * QT);
  int qi[QT];
  bool qvalid[QT];
  u16x8 qfrag[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    qi[t] = min(q0 + t * 16 + fi, N - 1);      // clamp: every lane stays active (transposed LDS reads need EXEC = all)
    qvalid[t] = (q0 + t * 16 + fi) < N;
    qfrag[t] = *reinterpret_cast<const u16x8*>(Qh + (seq0 + qi[t]) * 32 + fg * 8);
  }
  u16* Xh = X + (long long)h * Mtot * T * 32;
  const bool ragged = L != LP;                  // wave-uniform: pad keys need masking
  const bool active = q0 < N;                   // whole wave out of range (wave-uniform): it only helps staging
  for (int f0 = 0; f0 < T; f0 += TCH) {
  const int nf = min(TCH, T - f0);
  if (f0 > 0) __syncthreads();                  // every wave is done with the previous chunk
  // ---- stage K, V of frames f0 .. f0+nf-1 of this (sequence, head): 64-byte rows, 4 chunks of 16 B; pad keys are zero ----
  for (int c = tid; c < nf * LP * 4; c += nthreads) {
    int row = c >> 2, g = c & 3;
    int f = row / LP, l = row - f * LP;
    u16x8 kv = {0, 0, 0, 0, 0, 0, 0, 0}, vv = kv;
    if (l < L) {
      long long r = seq0 + (long long)(f0 + f) * L + l;
      kv = *reinterpret_cast<const u16x8*>(Kh + r * 32 + g * 8);
      vv = *reinterpret_cast<const u16x8*>(Vh + r * 32 + g * 8);
    }
    *reinterpret_cast<u16x8*>(sK + row * 32 + swz_chunk(row, g) * 8) = kv;
    *reinterpret_cast<u16x8*>(sV + v_lds_off(row, g * 8)) = vv;
  }
  __syncthreads();
  if (active)
  for (int fl = 0; fl < nf; ++fl) {
    const int f = f0 + fl;
    // S^T tiles: D[key][query]
    f32x4 sc[QT][2 * NKS];
#pragma unroll
    for (int kt = 0; kt < 2 * NKS; ++kt) {
      int row = fl * LP + kt * 16 + fi;
      u16x8 kf = *reinterpret_cast<const u16x8*>(sK + row * 32 + swz_chunk(row, fg) * 8);
#pragma unroll
      for (int t = 0; t < QT; ++t) sc[t][kt] = H16<BF>::mfma(kf, qfrag[t], f32x4{0.f, 0.f, 0.f, 0.f});
    }
    if (ragged) {
#pragma unroll
      for (int kt = 0; kt < 2 * NKS; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (kt * 16 + fg * 4 + r >= L) {
#pragma unroll
            for (int t = 0; t < QT; ++t) sc[t][kt][r] = -INFINITY;
          }
    }
    float inv[QT];
    u16x8 pf[QT][NKS];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      float mx = sc[t][0][0];
#pragma unroll
      for (int kt = 0; kt < 2 * NKS; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sc[t][kt][r]);
      mx = groups_max(mx);
      float sum = 0.f;
#pragma unroll
      for (int kt = 0; kt < 2 * NKS; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          sc[t][kt][r] = __builtin_amdgcn_exp2f(sc[t][kt][r] - mx);   // q carries scale*log2(e)
          sum += sc[t][kt][r];
        }
      inv[t] = 1.f / groups_sum(sum);

      if (attn != nullptr && qvalid[t]) {       // optional reference output space_attn[(s h), q, f, l]
        float* ap = attn + ((((long long)s * heads + h) * N + qi[t]) * T + f) * L;
#pragma unroll
        for (int kt = 0; kt < 2 * NKS; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            int key = kt * 16 + fg * 4 + r;
            if (key < L) ap[key] = sc[t][kt][r] * inv[t];
          }
      }
      // P^T fragments (B operand), unnormalised; the fp32 result is normalised instead
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[t][ks][j] = H16<BF>::from_f32(sc[t][2 * ks + (j >> 2)][j & 3]);
    }

    f32x4 xa[QT][2];
#pragma unroll
    for (int nd = 0; nd < 2; ++nd) {
#pragma unroll
      for (int t = 0; t < QT; ++t) xa[t][nd] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        // lane i of 16-lane group g supplies the address of row (i>>2), columns 4*(i&3).. of the 4x16 block
        int key0 = fl * LP + ks * 32 + fg * 4 + (fi >> 2);
        int dcol = nd * 16 + (fi & 3) * 4;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (s16x4 __attribute__((address_space(3)))*)(sV + v_lds_off(key0, dcol)));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (s16x4 __attribute__((address_space(3)))*)(sV + v_lds_off(key0 + 16, dcol)));
        u16x8 vf;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          vf[j] = (u16)lo[j];
          vf[4 + j] = (u16)hi[j];
        }
#pragma unroll
        for (int t = 0; t < QT; ++t) xa[t][nd] = H16<BF>::mfma(vf, pf[t][ks], xa[t][nd]);    // D[d][query]
      }
    }
    // lane holds channels nd*16 + 4g + r of its query: stored at position g*8 + nd*4 + r (perm32 order) -> 16 B per lane,
    // 1 KiB contiguous per wave store ([head][frame][row][32] layout, consecutive queries are consecutive rows)
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      if (qvalid[t]) {
        float v[8] = {xa[t][0][0] * inv[t], xa[t][0][1] * inv[t], xa[t][0][2] * inv[t], xa[t][0][3] * inv[t],
                      xa[t][1][0] * inv[t], xa[t][1][1] * inv[t], xa[t][1][2] * inv[t], xa[t][1][3] * inv[t]};
        long long row = (long long)f * Mtot + seq0 + qi[t];
        *reinterpret_cast<u16x8*>(Xh + row * 32 + fg * 8) = cvt8<BF>(v);
      }
    }
  }
  }
}

// ---- long frames (more than 256 keys per frame): full T*H*W trajectory attention (WC/temporal_attention.py:103-155 runs ONE
//      sequence of T*H*W tokens per clip, H*W keys per frame).  Same orientation and layouts as spatial_attn_kernel; a frame's keys
//      are visited in chunks of 256 (K / V of a chunk staged in LDS) with an online softmax per (query, frame): running maximum,
//      running denominator, rescaled accumulator -- the N x N logits (8.6 GB at 64 x 64 in the reference) never exist anywhere.
template <bool BF>
__global__ __launch_bounds__(512) void spatial_attn_long_kernel(const u16* __restrict__ Q16, const u16* __restrict__ K16,
                                                                const u16* __restrict__ V16, u16* __restrict__ X, int N, int T, int L,
                                                                int heads, long long Mtot) {
  extern __shared__ __attribute__((aligned(16))) u16 smem[];
  constexpr int NKS = 8, LP = NKS * 32, QT = 2;
  u16* sK = smem;
  u16* sV = smem + (size_t)LP * 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthreads = blockDim.x;
  const int h = blockIdx.y, s = blockIdx.z;
  const int fi = lane & 15, fg = lane >> 4;
  const long long seq0 = (long long)s * N;
  const u16* Kh = K16 + (long long)h * Mtot * 32;
  const u16* Vh = V16 + (long long)h * Mtot * 32;
  const u16* Qh = Q16 + (long long)h * Mtot * 32;
  const int q0 = (blockIdx.x * (nthreads >> 6) + wave) * (16 * QT);
  int qi[QT];
  bool qvalid[QT];
  u16x8 qfrag[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    qi[t] = min(q0 + t * 16 + fi, N - 1);
    qvalid[t] = (q0 + t * 16 + fi) < N;
    qfrag[t] = *reinterpret_cast<const u16x8*>(Qh + (seq0 + qi[t]) * 32 + fg * 8);
  }
  u16* Xh = X + (long long)h * Mtot * T * 32;
  const bool active = q0 < N;
  const int nch = (L + LP - 1) / LP;
  for (int f = 0; f < T; ++f) {
    float mrun[QT], srun[QT];
    f32x4 xa[QT][2];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      mrun[t] = -INFINITY;
      srun[t] = 0.f;
      xa[t][0] = xa[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int c = 0; c < nch; ++c) {
      const int l0 = c * LP, nk = min(LP, L - l0);           // keys of this chunk
      __syncthreads();                                       // every wave is done with the previous chunk
      for (int e = tid; e < LP * 4; e += nthreads) {
        const int row = e >> 2, g = e & 3;
        u16x8 kv = {0, 0, 0, 0, 0, 0, 0, 0}, vv = kv;
        if (row < nk) {
          const long long r = seq0 + (long long)f * L + l0 + row;
          kv = *reinterpret_cast<const u16x8*>(Kh + r * 32 + g * 8);
          vv = *reinterpret_cast<const u16x8*>(Vh + r * 32 + g * 8);
        }
        *reinterpret_cast<u16x8*>(sK + row * 32 + swz_chunk(row, g) * 8) = kv;
        *reinterpret_cast<u16x8*>(sV + v_lds_off(row, g * 8)) = vv;
      }
      __syncthreads();
      if (!active) continue;
      f32x4 sc[QT][2 * NKS];
#pragma unroll
      for (int kt = 0; kt < 2 * NKS; ++kt) {
        const int row = kt * 16 + fi;
        const u16x8 kf = *reinterpret_cast<const u16x8*>(sK + row * 32 + swz_chunk(row, fg) * 8);
#pragma unroll
        for (int t = 0; t < QT; ++t) sc[t][kt] = H16<BF>::mfma(kf, qfrag[t], f32x4{0.f, 0.f, 0.f, 0.f});
      }
      if (nk < LP) {
#pragma unroll
        for (int kt = 0; kt < 2 * NKS; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (kt * 16 + fg * 4 + r >= nk) {
#pragma unroll
              for (int t = 0; t < QT; ++t) sc[t][kt][r] = -INFINITY;
            }
      }
      u16x8 pf[QT][NKS];
#pragma unroll
      for (int t = 0; t < QT; ++t) {
        float mx = sc[t][0][0];
#pragma unroll
        for (int kt = 0; kt < 2 * NKS; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sc[t][kt][r]);
        mx = fmaxf(groups_max(mx), mrun[t]);
        const float alpha = __builtin_amdgcn_exp2f(mrun[t] - mx);            // 0 on the first chunk (mrun = -inf)
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2 * NKS; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            sc[t][kt][r] = __builtin_amdgcn_exp2f(sc[t][kt][r] - mx);
            sum += sc[t][kt][r];
          }
        srun[t] = srun[t] * alpha + groups_sum(sum);
        mrun[t] = mx;
        xa[t][0] *= alpha;
        xa[t][1] *= alpha;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
          for (int j = 0; j < 8; ++j) pf[t][ks][j] = H16<BF>::from_f32(sc[t][2 * ks + (j >> 2)][j & 3]);
      }
#pragma unroll
      for (int nd = 0; nd < 2; ++nd) {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          const int key0 = ks * 32 + fg * 4 + (fi >> 2);
          const int dcol = nd * 16 + (fi & 3) * 4;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(sV + v_lds_off(key0, dcol)));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(sV + v_lds_off(key0 + 16, dcol)));
          u16x8 vf;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            vf[j] = (u16)lo[j];
            vf[4 + j] = (u16)hi[j];
          }
#pragma unroll
          for (int t = 0; t < QT; ++t) xa[t][nd] = H16<BF>::mfma(vf, pf[t][ks], xa[t][nd]);
        }
      }
    }
    if (active) {
#pragma unroll
      for (int t = 0; t < QT; ++t) {
        if (qvalid[t]) {
          const float inv = 1.f / srun[t];
          float v[8] = {xa[t][0][0] * inv, xa[t][0][1] * inv, xa[t][0][2] * inv, xa[t][0][3] * inv,
                        xa[t][1][0] * inv, xa[t][1][1] * inv, xa[t][1][2] * inv, xa[t][1][3] * inv};
          const long long row = (long long)f * Mtot + seq0 + qi[t];
          *reinterpret_cast<u16x8*>(Xh + row * 32 + fg * 8) = cvt8<BF>(v);
        }
      }
    }
  }
}

}  // namespace axvs
