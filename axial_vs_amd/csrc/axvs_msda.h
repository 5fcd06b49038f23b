// Multi-scale deformable attention forward.
// Reference: OPS = MaXTron_Video-kMaX/maxtron_deeplab/modeling/within_clip_tracking_module/ops
//   module  OPS/modules/ms_deform_attn.py:81-125, core  OPS/functions/ms_deform_attn_func.py:55-77,
//   native op OPS/src/cuda/ms_deform_attn_cuda.cu:21-86 (im2col kernel: sample at loc*size - 0.5, zero outside).
// The op is a gather: per (query, head) L*P bilinear samples of a D-channel value row, so it is bound by L2/TA request
// rate, not by arithmetic; the layouts below make every sample corner one contiguous 64-byte (16-bit) segment.
#pragma once
#include "axvs_common.h"

namespace axvs {

constexpr int kMsdaMaxLevels = 8;
struct MsdaLevels {
  int H[kMsdaMaxLevels], W[kMsdaMaxLevels], start[kMsdaMaxLevels];
  int L;
};

// ---- compatibility core op: fp32 value [N,S,M,D] exactly as the reference extension takes it; one thread per output
//      element (n, q, m, d), d fastest (the D channels of a sample corner are D*4 contiguous bytes) ----
__global__ __launch_bounds__(256) void msda_core_kernel(const float* __restrict__ value, MsdaLevels lv,
                                                        const float* __restrict__ loc, const float* __restrict__ aw,
                                                        float* __restrict__ out, int N, int S, int M, int D, int Lq, int P) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)N * Lq * M * D;
  if (idx >= total) return;
  const int d = (int)(idx % D);
  long long r = idx / D;
  const int m = (int)(r % M);
  r /= M;                                                   // r = n*Lq + q
  const int n = (int)(r / Lq);
  const float* lp = loc + (r * M + m) * lv.L * P * 2;
  const float* ap = aw + (r * M + m) * lv.L * P;
  const float* vb = value + ((long long)n * S * M + m) * D + d;
  float acc = 0.f;
  for (int l = 0; l < lv.L; ++l) {
    const int H = lv.H[l], W = lv.W[l];
    const float* vl = vb + (long long)lv.start[l] * M * D;
    for (int p = 0; p < P; ++p) {
      const float x = lp[(l * P + p) * 2] * W - 0.5f, y = lp[(l * P + p) * 2 + 1] * H - 0.5f;
      const float a = ap[l * P + p];
      if (y > -1.f && x > -1.f && y < H && x < W) {
        const float xf = floorf(x), yf = floorf(y);
        const int x0 = (int)xf, y0 = (int)yf;
        const float fx = x - xf, fy = y - yf;
        float s = 0.f;
        if (y0 >= 0 && x0 >= 0) s += (1.f - fy) * (1.f - fx) * vl[((long long)y0 * W + x0) * M * D];
        if (y0 >= 0 && x0 + 1 < W) s += (1.f - fy) * fx * vl[((long long)y0 * W + x0 + 1) * M * D];
        if (y0 + 1 < H && x0 >= 0) s += fy * (1.f - fx) * vl[((long long)(y0 + 1) * W + x0) * M * D];
        if (y0 + 1 < H && x0 + 1 < W) s += fy * fx * vl[((long long)(y0 + 1) * W + x0 + 1) * M * D];
        acc += a * s;
      }
    }
  }
  out[idx] = acc;
}

// ---- backward of the compatibility core op (the reference extension's ms_deform_attn_backward: OPS/src/cuda/ms_deform_attn_cuda.cu:
//      89-157, kernels ms_deformable_col2im_* in OPS/src/cuda/ms_deform_im2col_cuda.cuh).  Same thread mapping as the forward: one
//      thread per (n, q, m, d), d fastest.  Per sample (l, p): the four corner values v1..v4 (zero outside the map), bilinear
//      weights w1 = hh hw, w2 = hh lw, w3 = lh hw, w4 = lh lw (lh = y - floor y, lw = x - floor x, hh = 1 - lh, hw = 1 - lw);
//        grad_value[corner][d]      += w_corner * attn * g[d]                  (atomic adds, like the reference)
//        grad_attn[l,p]              = sum_d g[d] (w1 v1 + w2 v2 + w3 v3 + w4 v4)
//        grad_loc[l,p] (x, y)        = sum_d attn g[d] (W (-hh v1 + hh v2 - lh v3 + lh v4),  H (-hw v1 - lw v2 + hw v3 + lw v4))
//      SHFL: D is a power of two <= 64: the sums over d are xor-shuffles inside the D lanes of a (n, q, m) group (fixed order);
//      otherwise atomic adds into zero-initialised grad_loc / grad_attn. ----
template <bool SHFL>
__global__ __launch_bounds__(256) void msda_core_bwd_kernel(const float* __restrict__ value, MsdaLevels lv, const float* __restrict__ loc,
                                                            const float* __restrict__ aw, const float* __restrict__ gout,
                                                            float* __restrict__ gvalue, float* __restrict__ gloc, float* __restrict__ gaw, int N,
                                                            int S, int M, int D, int Lq, int P) {
  const long long idx0 = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)N * Lq * M * D;
  const bool live = idx0 < total;
  const long long idx = live ? idx0 : total - 1;              // (dead lanes keep the shuffles of their group well defined)
  const int d = (int)(idx % D);
  long long r = idx / D;
  const int m = (int)(r % M);
  r /= M;                                                     // r = n*Lq + q
  const int n = (int)(r / Lq);
  const float* lp = loc + (r * M + m) * lv.L * P * 2;
  const float* ap = aw + (r * M + m) * lv.L * P;
  const long long vb = ((long long)n * S * M + m) * D + d;
  const float g = live ? gout[idx] : 0.f;
  for (int l = 0; l < lv.L; ++l) {
    const int H = lv.H[l], W = lv.W[l];
    const long long vl = vb + (long long)lv.start[l] * M * D;
    for (int p = 0; p < P; ++p) {
      const float x = lp[(l * P + p) * 2] * W - 0.5f, y = lp[(l * P + p) * 2 + 1] * H - 0.5f;
      const float a = ap[l * P + p];
      float ga = 0.f, gx = 0.f, gy = 0.f;
      if (y > -1.f && x > -1.f && y < H && x < W) {
        const float xf = floorf(x), yf = floorf(y);
        const int x0 = (int)xf, y0 = (int)yf;
        const float lw = x - xf, lh = y - yf, hw = 1.f - lw, hh = 1.f - lh;
        const float tg = g * a;
        float v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
        if (y0 >= 0 && x0 >= 0) {
          const long long o = vl + ((long long)y0 * W + x0) * M * D;
          v1 = value[o];
          if (live) atomicAdd(gvalue + o, hh * hw * tg);
        }
        if (y0 >= 0 && x0 + 1 < W) {
          const long long o = vl + ((long long)y0 * W + x0 + 1) * M * D;
          v2 = value[o];
          if (live) atomicAdd(gvalue + o, hh * lw * tg);
        }
        if (y0 + 1 < H && x0 >= 0) {
          const long long o = vl + ((long long)(y0 + 1) * W + x0) * M * D;
          v3 = value[o];
          if (live) atomicAdd(gvalue + o, lh * hw * tg);
        }
        if (y0 + 1 < H && x0 + 1 < W) {
          const long long o = vl + ((long long)(y0 + 1) * W + x0 + 1) * M * D;
          v4 = value[o];
          if (live) atomicAdd(gvalue + o, lh * lw * tg);
        }
        ga = g * (hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4);
        gx = W * tg * (-hh * v1 + hh * v2 - lh * v3 + lh * v4);
        gy = H * tg * (-hw * v1 - lw * v2 + hw * v3 + lw * v4);
      }
      const long long s = (r * M + m) * lv.L * P + l * P + p;
      if (SHFL) {
        for (int o = D >> 1; o > 0; o >>= 1) {
          ga += __shfl_xor(ga, o, 64);
          gx += __shfl_xor(gx, o, 64);
          gy += __shfl_xor(gy, o, 64);
        }
        if (live && d == 0) {
          gaw[s] = ga;
          gloc[2 * s] = gx;
          gloc[2 * s + 1] = gy;
        }
      } else if (live) {
        atomicAdd(gaw + s, ga);
        atomicAdd(gloc + 2 * s, gx);
        atomicAdd(gloc + 2 * s + 1, gy);
      }
    }
  }
}

// ---- module path: softmax over the L*P logits, sampling locations from the reference points, bilinear gather, weighted
//      sum.  value16: blocked 16-bit [M][N*S][32] (head blocks, written by the value_proj GEMM epilogue);
//      qproj: fp32 [N*Lq][3*M*L*P] = sampling offsets (M,L,P,2) | attention logits (M,L,P) (one GEMM on the query);
//      o16: blocked 16-bit [2M][N*Lq][32] for the output_proj GEMM: hi blocks then lo blocks (o = hi + lo, split precision).
//      4 lanes per (query, head): lane j owns channels 8j..8j+7 (16-byte loads), a sample corner is one 64-byte segment. ----
template <bool BF, int PT>   // PT > 0: n_points known at compile time (all 4*PT corner loads of a level in flight together)
__global__ __launch_bounds__(256) void msda_gather_kernel(const u16* __restrict__ value16, const float* __restrict__ qproj,
                                                          const float* __restrict__ refp, int ref_dim, MsdaLevels lv,
                                                          u16* __restrict__ o16, int N, int S, int Lq, int M, int Prt,
                                                          float* __restrict__ of32 = nullptr, int d = 32) {
  // of32 != nullptr: the sampled rows go out as fp32 [N*Lq][M*d] (natural channel order) instead of the split 16-bit blocks
  // -- the Tube-Link plugin runs its temporal encoder on them before output_proj (TL ...pixel_decoder.py:613-633).
  const int P = PT > 0 ? PT : Prt;
  const long long gid = (long long)blockIdx.x * 64 + (threadIdx.x >> 2);        // (row, head) group
  const int j = threadIdx.x & 3;
  const long long R = (long long)N * Lq;
  const bool valid = gid < R * M;
  const long long g = valid ? gid : R * M - 1;
  const int m = (int)(g % M);
  const long long row = g / M;
  const int n = (int)(row / Lq);
  const int L = lv.L, LP = L * P, MLP = M * LP;
  const float* offs = qproj + row * 3 * MLP + (long long)m * LP * 2;
  const float* logit = qproj + row * 3 * MLP + 2 * MLP + (long long)m * LP;
  float mx = -INFINITY;
  for (int i = 0; i < LP; ++i) mx = fmaxf(mx, logit[i]);
  float den = 0.f;
  for (int i = 0; i < LP; ++i) den += __expf(logit[i] - mx);
  const float inv = 1.f / den;
  const u16* vbase = value16 + ((long long)m * N * S + (long long)n * S) * 32 + j * 8;
  const float* rp = refp + row * L * ref_dim;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int l = 0; l < L; ++l) {
    const int H = lv.H[l], W = lv.W[l];
    const u16* vl = vbase + (long long)lv.start[l] * 32;
    const float rx = rp[l * ref_dim], ry = rp[l * ref_dim + 1];
    float sx, sy;                                                    // offset scale (modules/ms_deform_attn.py:107-113)
    if (ref_dim == 2) { sx = 1.f / W; sy = 1.f / H; }
    else { sx = rp[l * ref_dim + 2] * 0.5f / P; sy = rp[l * ref_dim + 3] * 0.5f / P; }
#pragma unroll 4
    for (int p = 0; p < (PT > 0 ? PT : P); ++p) {
      const int i = l * P + p;
      const float a = __expf(logit[i] - mx) * inv;
      const float x = (rx + offs[i * 2] * sx) * W - 0.5f, y = (ry + offs[i * 2 + 1] * sy) * H - 0.5f;
      // branch-free: corners outside the map get weight 0 and a clamped (valid) address, so every load of the level can be
      // in flight at once
      const float xf = floorf(x), yf = floorf(y);
      const float fx = x - xf, fy = y - yf;
      const int x0 = (int)fmaxf(fminf(xf, (float)W), -2.f), y0 = (int)fmaxf(fminf(yf, (float)H), -2.f);
      const bool t = y0 >= 0 && y0 < H, b = y0 + 1 >= 0 && y0 + 1 < H, lft = x0 >= 0 && x0 < W, rgt = x0 + 1 >= 0 && x0 + 1 < W;
      const int yc0 = min(max(y0, 0), H - 1), yc1 = min(max(y0 + 1, 0), H - 1), xc0 = min(max(x0, 0), W - 1), xc1 = min(max(x0 + 1, 0), W - 1);
      const u16x8 v00 = *reinterpret_cast<const u16x8*>(vl + ((long long)yc0 * W + xc0) * 32);
      const u16x8 v01 = *reinterpret_cast<const u16x8*>(vl + ((long long)yc0 * W + xc1) * 32);
      const u16x8 v10 = *reinterpret_cast<const u16x8*>(vl + ((long long)yc1 * W + xc0) * 32);
      const u16x8 v11 = *reinterpret_cast<const u16x8*>(vl + ((long long)yc1 * W + xc1) * 32);
      const float w00 = (t && lft) ? a * (1.f - fy) * (1.f - fx) : 0.f, w01 = (t && rgt) ? a * (1.f - fy) * fx : 0.f;
      const float w10 = (b && lft) ? a * fy * (1.f - fx) : 0.f, w11 = (b && rgt) ? a * fy * fx : 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c)
        acc[c] += w00 * H16<BF>::to_f32(v00[c]) + w01 * H16<BF>::to_f32(v01[c]) + w10 * H16<BF>::to_f32(v10[c]) +
                  w11 * H16<BF>::to_f32(v11[c]);
    }
  }
  if (valid && of32) {
    if (j * 8 < d) {
      float* o = of32 + row * (long long)(M * d) + m * d + j * 8;
      *reinterpret_cast<float4*>(o) = float4{acc[0], acc[1], acc[2], acc[3]};
      *reinterpret_cast<float4*>(o + 4) = float4{acc[4], acc[5], acc[6], acc[7]};
    }
  } else if (valid) {
    const u16x8 hi = cvt8<BF>(acc);
    float lo[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) lo[c] = acc[c] - H16<BF>::to_f32(hi[c]);
    *reinterpret_cast<u16x8*>(o16 + ((long long)m * R + row) * 32 + j * 8) = hi;
    *reinterpret_cast<u16x8*>(o16 + ((long long)(M + m) * R + row) * 32 + j * 8) = cvt8<BF>(lo);
  }
}

}  // namespace axvs
