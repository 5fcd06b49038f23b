// Cross-clip tracking module kernels (CC/maxtron_cross_clip_tracking_module.py): everything that is not a plain GEMM.
#pragma once
#include "axvs_common.h"

namespace axvs {

// Temporal ASPP folded at pack time (CC/maxtron_cross_clip_tracking_module.py:176-201).  The three dilated Conv1d branches (bias, no
// activation, no norm) feed the bias-free 1x1 projection directly, so branches + concat + projection are ONE linear map of the
// clip axis:   y[t] = sum_j M_j x[clamp(t + off_j)] + b',   M_0 = sum_r P_r W_r[:,:,1] (the three centre taps, offset 0),
// M_{1+2r} = P_r W_r[:,:,0] (offset -rate_r), M_{2+2r} = P_r W_r[:,:,2] (offset +rate_r), b' = sum_r P_r b_r, P_r = proj[:, 256r..256r+255].
// 7 x 256 x 256 weights instead of 9 x 256 x 256 + 256 x 768, one GEMM with K = 1792 instead of three (K = 768) plus one (K = 768), no
// [rows, 768] intermediate, and one 16-bit rounding of an intermediate less.  The products are formed in fp32 (<= 768 terms) and rounded
// once to the operand type.  The rates enter at run time as the row offsets only (the packed buffer does not depend on them).
// thread -> (j, n, c); out: blocked weight layout (wblk_off) [K = 7 * 256][256 rows]
template <bool BF>
__global__ void pack_aspp_taps_kernel(const float* __restrict__ W0, const float* __restrict__ W1, const float* __restrict__ W2,
                                      const float* __restrict__ P /* [256][768] */, u16* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 7 * 256 * 256) return;
  const int c = idx & 255, n = (idx >> 8) & 255, j = idx >> 16;
  const float* Wr[3] = {W0, W1, W2};
  float acc = 0.f;
  if (j == 0) {
    for (int r = 0; r < 3; ++r)
      for (int m = 0; m < 256; ++m) acc += P[n * 768 + r * 256 + m] * Wr[r][(m * 256 + c) * 3 + 1];
  } else {
    const int r = (j - 1) >> 1, tap = ((j - 1) & 1) * 2;
    for (int m = 0; m < 256; ++m) acc += P[n * 768 + r * 256 + m] * Wr[r][(m * 256 + c) * 3 + tap];
  }
  out[wblk_off(256, n, j * 256 + c)] = H16<BF>::from_f32(acc);
}
__global__ void pack_aspp_bias_kernel(const float* __restrict__ b0, const float* __restrict__ b1, const float* __restrict__ b2,
                                      const float* __restrict__ P, float* __restrict__ out) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= 256) return;
  const float* br[3] = {b0, b1, b2};
  float acc = 0.f;
  for (int r = 0; r < 3; ++r)
    for (int m = 0; m < 256; ++m) acc += P[n * 768 + r * 256 + m] * br[r][m];
  out[n] = acc;
}

// eval-mode BatchNorm folded into a per-channel multiplier / bias:  y = (x - mean) / sqrt(var + eps) * w + b
__global__ void bn_fold_kernel(const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ mean,
                               const float* __restrict__ var, float eps, float* __restrict__ mul, float* __restrict__ add, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = w[i] * rsqrtf(var[i] + eps);
  mul[i] = s;
  add[i] = b[i] - mean[i] * s;
}

// ASPP tail + block residual (CC/...:186-201, :293-295), one wave per (bq, t) row, C = 256:
//   z = GELU(LN_cf(y; eps 1e-6) * g_a + b_a);  out = LN(z + x; eps 1e-5) * g_n + b_n
// affine != 0 (norm_fn 'syncbn' in eval mode / 'none': kmax_pixel_decoder.py:32-40): z = GELU(y * g_a + b_a) with g_a, b_a the folded running statistics
__global__ __launch_bounds__(256) void cc_aspp_post_kernel(const float* __restrict__ Y, const float* __restrict__ Xin,
                                                           const float* __restrict__ ga, const float* __restrict__ ba,
                                                           const float* __restrict__ gn, const float* __restrict__ bn,
                                                           float* __restrict__ out, long long M,
                                                           float* __restrict__ out2 = nullptr /* optional second copy of the rows */, int affine = 0) {
  constexpr int C = 256;
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float4 y = *reinterpret_cast<const float4*>(Y + row * C + lane * 4);
  const float4 x = *reinterpret_cast<const float4*>(Xin + row * C + lane * 4);
  float d0 = y.x, d1 = y.y, d2 = y.z, d3 = y.w, rstd = 1.f;
  if (!affine) {
    const float mu = wave_sum(y.x + y.y + y.z + y.w) * (1.f / C);
    d0 = y.x - mu; d1 = y.y - mu; d2 = y.z - mu; d3 = y.w - mu;
    rstd = 1.f / sqrtf(wave_sum(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3) * (1.f / C) + 1e-6f);
  }
  const float4 g = *reinterpret_cast<const float4*>(ga + lane * 4), b = *reinterpret_cast<const float4*>(ba + lane * 4);
  auto gelu = [](float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752f)); };
  const float z0 = gelu(d0 * rstd * g.x + b.x) + x.x, z1 = gelu(d1 * rstd * g.y + b.y) + x.y;
  const float z2 = gelu(d2 * rstd * g.z + b.z) + x.z, z3 = gelu(d3 * rstd * g.w + b.w) + x.w;
  const float mu2 = wave_sum(z0 + z1 + z2 + z3) * (1.f / C);
  d0 = z0 - mu2; d1 = z1 - mu2; d2 = z2 - mu2; d3 = z3 - mu2;
  const float rstd2 = rsqrtf(wave_sum(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3) * (1.f / C) + 1e-5f);
  const float4 g2 = *reinterpret_cast<const float4*>(gn + lane * 4), b2 = *reinterpret_cast<const float4*>(bn + lane * 4);
  const float4 o = float4{d0 * rstd2 * g2.x + b2.x, d1 * rstd2 * g2.y + b2.y, d2 * rstd2 * g2.z + b2.z, d3 * rstd2 * g2.w + b2.w};
  *reinterpret_cast<float4*>(out + row * C + lane * 4) = o;
  if (out2) *reinterpret_cast<float4*>(out2 + row * C + lane * 4) = o;
}

// [K1][256] fp32 -> [256][K1]: the class heads read their weight matrix with the CLASS index on the lanes (coalesced rows of K1 floats)
__global__ void transpose_k1x256_kernel(const float* __restrict__ src, float* __restrict__ dst, int K1) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= K1 * 256) return;
  const int k = i >> 8, c = i & 255;
  dst[(long long)c * K1 + k] = src[i];
}

// logits[k] = bc[k] + sum_c wcT[c][k] * pooled[c] (+ extra on the last class), 256 threads: thread (k & 127, channel half).  Round 5: the weight rows used to
// be read with the class index on the threads and a 1-KiB stride between them (12 us for the four layers of BASELINE config 4; now coalesced rows).
__device__ __forceinline__ void class_logits_256(const float* __restrict__ wcT, const float* __restrict__ bc, const float* pooled /* LDS, [256] */,
                                                 float* part /* LDS, [256] */, float* __restrict__ out, int K1, float last_extra, int tid) {
  const int kk = tid & 127, h = tid >> 7;
  for (int k0 = 0; k0 < K1; k0 += 128) {
    const int k = k0 + kk;
    float acc = 0.f;
    if (k < K1) {
      const float* w = wcT + (long long)(h * 128) * K1 + k;
      // 32 independent row loads in flight per thread: with 8 the loop was 16 dependent L2 round trips (the kernel's whole run time)
#pragma unroll 1
      for (int c0 = 0; c0 < 128; c0 += 32) {
        float wv[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) wv[j] = w[(long long)(c0 + j) * K1];
#pragma unroll
        for (int j = 0; j < 32; ++j) acc += wv[j] * pooled[h * 128 + c0 + j];
      }
    }
    part[tid] = acc;
    __syncthreads();
    if (h == 0 && k < K1) out[k] = bc[k] + (part[kk] + part[128 + kk]) + (k == K1 - 1 ? last_extra : 0.f);
    __syncthreads();
  }
}

// Class head of MaXTronCCPredictor (CC/...:48-52): per query q, softmax over ALL (b, clip) entries of a 256->1 activation
// head, weighted sum of the class embeddings, 256->K1 class head, void bias on the last class.
// emb: fp32 [(b q t)][ld] (class embedding = columns 0..255); out: fp32 [Q][K1].  One workgroup (256 threads) per q.
__global__ __launch_bounds__(256) void cc_class_head_kernel(const float* __restrict__ emb, int ld, const float* __restrict__ wa,
                                                            const float* __restrict__ ba, const float* __restrict__ wc /* transposed: [256][K1] */,
                                                            const float* __restrict__ bc, float* __restrict__ out, int Bv, int Q,
                                                            int Tc, int K1, float void_bias) {
  constexpr int C = 256, MAXE = 1024;
  __shared__ float logit[MAXE], pooled[C], red[16];
  const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  emb += (long long)blockIdx.y * Bv * Q * Tc * ld;    // blockIdx.y: layer (the module loop runs the heads of all layers in one launch)
  out += (long long)blockIdx.y * Q * K1;
  const int E = Bv * Tc;                              // entries the softmax runs over (dim 0 of the reference tensor)
  const float wac = wa[tid];
  auto row = [&](int e) { const int b = e / Tc, t = e - b * Tc; return emb + (((long long)b * Q + q) * Tc + t) * ld + tid; };
  // entries in groups of 4: their loads are independent (one L2 round trip per group instead of one per entry), one barrier pair per group
  for (int e0 = 0; e0 < E; e0 += 4) {
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = *row(min(e0 + j, E - 1)) * wac;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float sj = wave_sum(v[j]);
      if (lane == 0) red[j * 4 + wave] = sj;
    }
    __syncthreads();
    if (tid < 4 && e0 + tid < E) logit[e0 + tid] = red[tid * 4] + red[tid * 4 + 1] + red[tid * 4 + 2] + red[tid * 4 + 3] + ba[0];
    __syncthreads();
  }
  float mx = -INFINITY;
  for (int e = 0; e < E; ++e) mx = fmaxf(mx, logit[e]);
  float sum = 0.f;
  for (int e = 0; e < E; ++e) sum += __expf(logit[e] - mx);
  float p = 0.f;
  for (int e0 = 0; e0 < E; e0 += 4) {
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = *row(min(e0 + j, E - 1));
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (e0 + j < E) p += __expf(logit[e0 + j] - mx) / sum * v[j];
  }
  pooled[tid] = p;
  __syncthreads();
  class_logits_256(wc, bc, pooled, logit /* free now (>= 256 floats) */, out + (long long)q * K1, K1, void_bias, tid);
}

// Mask logits as a strided batched contraction over channels, shared by the two heads:
//   out[b, q, unit u, p] = mul * sum_c kern[(b q clip(u)), c] * feat[b, u, c, p] + add,        p = 0..P-1 pixels of a unit
// Video-kMaX (CC/...:61-69): unit = clip, P = V*H*W, feat = panoptic_features [B,CK,(Tc P)], out [B,Q,(Tc P)], clip(u) = u,
//   mul/add = eval BatchNorm(1).   Tube-Link (TLCC:774-778): unit = frame, P = h*w, feat = mask_feature [B,T,CK,P],
//   out [B,T,Q,P], clip(u) = u / frames_per_clip, mul/add = 1/0.
// Workgroup = 64 pixels of one unit (4 waves x 16 pixels) x all queries; pixels are the MFMA rows (A operand, transposed
// into LDS from the channels-first fp32 feature), queries the columns, so a lane stores 4 consecutive pixels of one query.
struct EinsumMap {
  long long f_b, f_u, f_c;     // feature strides (elements): batch, unit, channel
  long long o_b, o_u, o_q;     // output strides: batch, unit, query
  int units, upc;              // units per batch entry; units per clip
};

// GEN: pixel rows that start at any 4-byte boundary / P % 4 != 0 (193 x 337 maps) -- its own instantiation: the run-time
// alignment branches in the loads cost the common case 18 % (241 -> 285 us at config 4)
template <bool BF, int CK, bool GEN = false>
__global__ __launch_bounds__(256) void mask_einsum_kernel(const float* __restrict__ pf, const u16* __restrict__ kern16,
                                                          float* __restrict__ out, int Q, int Tc, long long P, long long Rk,
                                                          EinsumMap mp, const float* __restrict__ pix_bn /* {mul, add} or null */,
                                                          int nl = 1 /* layers sharing the staged feature tile */,
                                                          long long kstride = 0 /* elements between the layers' kernels */,
                                                          long long ostride = 0 /* ... and between their outputs */,
                                                          int al = 4 /* alignment (floats) of the pixel rows of pf and out: P % 4 != 0 -> 2 or 1 */) {
  // Workgroup = 256 pixels of one unit (4 waves x 64 pixels = 4 MFMA row tiles per wave): every kernel fragment fetched from L2
  // feeds 4 MFMAs, and a wave's stores cover 256 contiguous bytes of a query's row (round 2; 64 pixels per workgroup before: the
  // kernel fragments were 3x the traffic of the features).
  constexpr int KB = CK / 32, PXW = 4, PXT = 64 * PXW;      // px tiles per wave, pixels per workgroup
  extern __shared__ __attribute__((aligned(16))) u16 spx[];   // [kb][pixel][32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fi = lane & 15, fg = lane >> 4;
  const int bu = blockIdx.y, b = bu / mp.units, u = bu - b * mp.units, clip = u / mp.upc;
  const long long p0 = (long long)blockIdx.x * PXT;
  const float bn_mul = pix_bn ? pix_bn[0] : 1.f, bn_add = pix_bn ? pix_bn[1] : 0.f;
  // stage: thread -> (channel c, 32-pixel segment); 8 float4 loads of 128 contiguous bytes
  for (int idx = tid; idx < CK * (PXT / 32); idx += 256) {
    const int c = idx / (PXT / 32), seg = idx - c * (PXT / 32);
    const float* src = pf + b * mp.f_b + u * mp.f_u + c * mp.f_c + p0 + seg * 32;
    float4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const long long p = p0 + seg * 32 + i * 4;
      if constexpr (GEN) v[i] = ldg4(src + i * 4, (int)(P - p < 4 ? (P - p < 0 ? 0 : P - p) : 4), al);
      else v[i] = p + 3 < P ? *reinterpret_cast<const float4*>(src + i * 4) : float4{0.f, 0.f, 0.f, 0.f};
    }
    const int kb = c >> 5, k = c & 31;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float e[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int px = seg * 32 + i * 4 + j;
        spx[(kb * PXT + px) * 32 + swz_chunk(px, k >> 3) * 8 + (k & 7)] = H16<BF>::from_f32(e[j]);
      }
      if ((i & 1) == 1) lds_fence();
    }
  }
  __syncthreads();
  u16x8 af[PXW][KB];
#pragma unroll
  for (int t = 0; t < PXW; ++t)
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const int px = wave * (16 * PXW) + t * 16 + fi;
      af[t][kb] = *reinterpret_cast<const u16x8*>(spx + (kb * PXT + px) * 32 + swz_chunk(px, fg) * 8);
    }
  // The feature tile is staged ONCE and contracted with the mask kernels of every layer: the per-layer predictions of the
  // cross-clip modules all read the same pixel features (4 layers: 33.5 MB read once instead of four times)
  for (int ly = 0; ly < nl; ++ly) {
    const u16* kl = kern16 + ly * kstride;
    float* ol = out + ly * ostride;
    for (int qt = 0; qt * 16 < Q; ++qt) {
      const int q = min(qt * 16 + fi, Q - 1);
      const long long r = ((long long)b * Q + q) * Tc + clip;
      u16x8 bf[KB];
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) bf[kb] = *reinterpret_cast<const u16x8*>(kl + ((long long)kb * Rk + r) * 32 + fg * 8);
#pragma unroll
      for (int t = 0; t < PXW; ++t) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) acc = H16<BF>::mfma(af[t][kb], bf[kb], acc);          // D[pixel][query]
        const long long p = p0 + wave * (16 * PXW) + t * 16 + fg * 4;
        const float4 o4 = float4{acc[0] * bn_mul + bn_add, acc[1] * bn_mul + bn_add, acc[2] * bn_mul + bn_add, acc[3] * bn_mul + bn_add};
        if constexpr (GEN) {
          if (qt * 16 + fi < Q && p < P) stg4(ol + b * mp.o_b + u * mp.o_u + q * mp.o_q + p, o4, (int)(P - p < 4 ? P - p : 4), al);
        } else {
          if (qt * 16 + fi < Q && p + 3 < P) *reinterpret_cast<float4*>(ol + b * mp.o_b + u * mp.o_u + q * mp.o_q + p) = o4;      // (nontemporal stores: +2 us at config 4, round 5)
        }
      }
    }
  }
}
constexpr int kEinsumPx = 256;   // pixels per workgroup of mask_einsum_kernel
template <int CK>
constexpr size_t einsum_lds_bytes() { return (size_t)(CK / 32) * kEinsumPx * 32 * sizeof(u16); }

// Tube-Link class head (TLCC:783-797), one workgroup (256 threads = channels) per (b, q):
//   a_t = softmax_t(w_a . x[b,q,t,:] + b_a);  pooled = sum_t a_t x[b,q,t,:];  logits = W_c pooled + b_c.
// x: fp32 [(b q t)][256] (post-normed); out: fp32 [(b q)][K1].
__global__ __launch_bounds__(256) void tl_class_head_kernel(const float* __restrict__ x, const float* __restrict__ wa,
                                                            const float* __restrict__ ba, const float* __restrict__ wc,
                                                            const float* __restrict__ bc, float* __restrict__ out, int Tc, int K1) {
  constexpr int C = 256, MAXT = 1024;
  __shared__ float logit[MAXT], pooled[C], red[16];
  const int bq = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* xr = x + ((long long)blockIdx.y * gridDim.x + bq) * Tc * C;   // blockIdx.y: layer
  out += (long long)blockIdx.y * gridDim.x * K1;
  const float wac = wa[tid];
  for (int t0 = 0; t0 < Tc; t0 += 4) {      // clips in groups of 4 (independent loads, one barrier pair per group)
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = xr[min(t0 + j, Tc - 1) * C + tid] * wac;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float sj = wave_sum(v[j]);
      if (lane == 0) red[j * 4 + wave] = sj;
    }
    __syncthreads();
    if (tid < 4 && t0 + tid < Tc) logit[t0 + tid] = red[tid * 4] + red[tid * 4 + 1] + red[tid * 4 + 2] + red[tid * 4 + 3] + ba[0];
    __syncthreads();
  }
  float mx = -INFINITY;
  for (int t = 0; t < Tc; ++t) mx = fmaxf(mx, logit[t]);
  float sum = 0.f;
  for (int t = 0; t < Tc; ++t) sum += __expf(logit[t] - mx);
  float p = 0.f;
  for (int t0 = 0; t0 < Tc; t0 += 4) {
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = xr[min(t0 + j, Tc - 1) * C + tid];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (t0 + j < Tc) p += __expf(logit[t0 + j] - mx) / sum * v[j];
  }
  pooled[tid] = p;
  __syncthreads();
  class_logits_256(wc, bc, pooled, logit /* free now (>= 256 floats) */, out + (long long)bq * K1, K1, 0.f, tid);
}

}  // namespace axvs
