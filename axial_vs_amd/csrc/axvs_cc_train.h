// Kernels of the cross-clip tracking module's TRAINING tier (forward that keeps activations + backward), next to the trajectory
// attention / LayerNorm / GEMM kernels it shares with the within-clip layer (axvs_train.h, axvs_train_gemm.h).
//
// Reference: CC = MaXTron_Video-kMaX/maxtron_deeplab/modeling/cross_clip_tracking_module/maxtron_cross_clip_tracking_module.py
//   ASPP (CC:176-201): three Conv1d(256,256,3, dilation r, padding 'same', replicate) over the clip axis, concatenated, 1x1
//   projection, channels-first LayerNorm (eps 1e-6), GELU, dropout; here the convolutions are GEMMs over an im2col of the rows.
//   ConvBN heads (CC:266-270, :33-43) with (Sync)BatchNorm in train mode: batch statistics over all rows of one layer's call
//   (eps 1e-3), summed over ranks by the caller's all-reduce between the statistics kernel and the apply kernel.
//   MaXTronCCPredictor.forward, training branch (CC:45-57): class-activation softmax over the (b t) axis, pooled class
//   embedding, class head, mask head, 'bchw,bcn->bnhw' einsum, one-channel BatchNorm on the mask logits.
// Activations are fp32 rows [rows][C]; the row order of a layer is the reference's clip_query layout (b, q, t).
#pragma once
#include "axvs_train.h"

namespace axvs {
namespace tr {

__device__ __forceinline__ float gelu_f(float z) { return 0.5f * z * (1.f + erff(z * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_d(float z) {   // d gelu / dz
  return 0.5f * (1.f + erff(z * 0.70710678118654752f)) + z * 0.39894228040143268f * expf(-0.5f * z * z);
}

// ---- ASPP: im2col over the clip axis ---------------------------------------------------------------------------------------
// xcol[m][c*3 + j] = y[m - t + clamp(t + (j-1) rate, 0, Tc-1)][c], t = m % Tc: the Conv1d weight [Cout][C][3] is then the
// row-major [Cout][3C] operand of a plain GEMM (replicate padding = the clamp).
__global__ __launch_bounds__(256) void cct_im2col_kernel(const float* __restrict__ y, float* __restrict__ xcol, long long M, int Tc, int C,
                                                          int rate) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int K3 = 3 * C;
  if (i >= (size_t)M * K3) return;
  const long long m = (long long)(i / K3);
  const int e = (int)(i - (size_t)m * K3), c = e / 3, j = e - 3 * c;
  const int t = (int)(m % Tc);
  int tt = t + (j - 1) * rate;
  tt = tt < 0 ? 0 : (tt > Tc - 1 ? Tc - 1 : tt);
  xcol[i] = y[(m - t + tt) * C + c];
}

// dy[m][c] += sum over taps j and source positions t' with clamp(t' + (j-1) rate) == t of dxcol[m - t + t'][c*3 + j]: the gather
// form of the im2col's transpose for one branch (deterministic, no atomics)
__global__ __launch_bounds__(256) void cct_col2im_add_kernel(const float* __restrict__ dxcol, float* __restrict__ dy, long long M, int Tc, int C,
                                                              int rate) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)M * C) return;
  const long long m = (long long)(i / C);
  const int c = (int)(i - (size_t)m * C);
  const int t = (int)(m % Tc);
  const size_t K3 = (size_t)3 * C;
  float acc = 0.f;
  for (int j = 0; j < 3; ++j)
    for (int ts = 0; ts < Tc; ++ts) {
      int tt = ts + (j - 1) * rate;
      tt = tt < 0 ? 0 : (tt > Tc - 1 ? Tc - 1 : tt);
      if (tt == t) acc += dxcol[(size_t)(m - t + ts) * K3 + c * 3 + j];
    }
  dy[i] += acc;
}

// u[m][c] = y[m][c] + keep * gelu(z[m][c]); the dropout index is the element index of the reference's [(b q), C, Tc] tensor
// (CC:197-199): ((m / Tc) C + c) Tc + m % Tc
__global__ __launch_bounds__(256) void cct_gelu_drop_res_kernel(const float* __restrict__ z, const float* __restrict__ y, float* __restrict__ u,
                                                                 long long M, int Tc, int C, Drop dr) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)M * C) return;
  const long long m = (long long)(i / C);
  const int c = (int)(i - (size_t)m * C);
  const unsigned long long e = ((unsigned long long)(m / Tc) * C + c) * Tc + (unsigned long long)(m % Tc);
  u[i] = y[i] + drop_keep(dr, e) * gelu_f(z[i]);
}

// dz = keep * gelu'(z) * du;  dy = du (the residual branch of u = y + ...)
__global__ __launch_bounds__(256) void cct_gelu_drop_bwd_kernel(const float* __restrict__ du, const float* __restrict__ z, float* __restrict__ dz,
                                                                 float* __restrict__ dy, long long M, int Tc, int C, Drop dr) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)M * C) return;
  const long long m = (long long)(i / C);
  const int c = (int)(i - (size_t)m * C);
  const unsigned long long e = ((unsigned long long)(m / Tc) * C + c) * Tc + (unsigned long long)(m % Tc);
  const float d = du[i];
  dy[i] = d;
  dz[i] = d * drop_keep(dr, e) * gelu_d(z[i]);
}

// ---- BatchNorm with batch statistics, G groups (= layers: one statistics set per call of the reference module) of R rows -------
// stage 1: part[(g nblk + blk)][2][C] = sum over the block's rows of (x - s_c), (x - s_c)^2.  The shift s = the running mean
// (identical on every rank) keeps E[d^2] - E[d]^2 well conditioned; nullptr: 0.
__global__ __launch_bounds__(256) void cct_bn_stats_kernel(const float* __restrict__ x, const float* __restrict__ shift, float* __restrict__ part,
                                                            long long R, int C, int rows_per_blk) {
  const int g = blockIdx.y, nblk = gridDim.x;
  const long long r0 = (long long)blockIdx.x * rows_per_blk, r1 = r0 + rows_per_blk < R ? r0 + rows_per_blk : R;
  const float* xg = x + (size_t)g * R * C;
  float* out = part + ((size_t)g * nblk + blockIdx.x) * 2 * C;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float s = shift ? shift[c] : 0.f;
    float a = 0.f, b = 0.f;
    for (long long r = r0; r < r1; ++r) {
      const float d = xg[r * C + c] - s;
      a += d;
      b += d * d;
    }
    out[c] = a;
    out[C + c] = b;
  }
}

// out[g][i] = sum_blk part[(g nblk + blk) n + i] (deterministic); grid (ceil(n / 4), G).  out2 (nullable): a
// second copy of the sums (the all-reduce buffer next to this rank's own sums); count_ptr (nullable): receives count_val (the row
// count that travels with the sums through the all-reduce)
__global__ __launch_bounds__(256) void cct_reduce_groups_kernel(const float* __restrict__ part, int nblk, int n, float* __restrict__ out,
                                                                 float* __restrict__ out2, float* __restrict__ count_ptr, float count_val) {
  // one wave per output element: lanes stride over the partials (double accumulation), xor-shuffle tree -- a fixed order
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, g = blockIdx.y;
  if (count_ptr && blockIdx.x == 0 && threadIdx.x == 0 && g == 0) *count_ptr = count_val;
  if (i >= n) return;
  double a = 0.0;
  for (int b = lane; b < nblk; b += 64) a += (double)part[((size_t)g * nblk + b) * n + i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
  if (lane == 0) {
    out[(size_t)g * n + i] = (float)a;
    if (out2) out2[(size_t)g * n + i] = (float)a;
  }
}

__global__ __launch_bounds__(256) void cct_add_inplace_kernel(float* __restrict__ y, const float* __restrict__ a, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] += a[i];
}

// sums: [G][2][C] (possibly summed over ranks), *count: rows that went into them.  mean / rstd: [G][C]; stats_out (nullable):
// [G][2][C] = batch mean and UNBIASED variance (what the running statistics are updated with, momentum on the caller's side)
__global__ __launch_bounds__(256) void cct_bn_finalize_kernel(const float* __restrict__ sums, const float* __restrict__ count,
                                                               const float* __restrict__ shift, float eps, float* __restrict__ mean,
                                                               float* __restrict__ rstd, float* __restrict__ stats_out, int C) {
  const int c = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y;
  if (c >= C) return;
  const double n = (double)*count;
  const double m1 = (double)sums[((size_t)g * 2) * C + c] / n, m2 = (double)sums[((size_t)g * 2 + 1) * C + c] / n;
  double var = m2 - m1 * m1;
  var = var > 0.0 ? var : 0.0;
  const double mu = (shift ? (double)shift[c] : 0.0) + m1;
  mean[(size_t)g * C + c] = (float)mu;
  rstd[(size_t)g * C + c] = (float)(1.0 / sqrt(var + (double)eps));
  if (stats_out) {
    stats_out[((size_t)g * 2) * C + c] = (float)mu;
    stats_out[((size_t)g * 2 + 1) * C + c] = (float)(n > 1.0 ? var * n / (n - 1.0) : var);
  }
}

// out = act(xhat w + b), xhat = (x - mean_g) rstd_g
__global__ __launch_bounds__(256) void cct_bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ out,
                                                            long long R, int C, int G, int gelu) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)G * R * C) return;
  const int c = (int)(i % C);
  const int g = (int)(i / ((size_t)R * C));
  const float v = (x[i] - mean[(size_t)g * C + c]) * rstd[(size_t)g * C + c] * w[c] + b[c];
  out[i] = gelu ? gelu_f(v) : v;
}

// backward statistics: dz = gelu ? dy gelu'(xhat w + b) : dy;  part[(g nblk + blk)][2][C] = sum dz, sum dz xhat
__global__ __launch_bounds__(256) void cct_bn_bwd_stats_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ part,
                                                                long long R, int C, int rows_per_blk, int gelu) {
  const int g = blockIdx.y, nblk = gridDim.x;
  const long long r0 = (long long)blockIdx.x * rows_per_blk, r1 = r0 + rows_per_blk < R ? r0 + rows_per_blk : R;
  const size_t base = (size_t)g * R * C;
  float* out = part + ((size_t)g * nblk + blockIdx.x) * 2 * C;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float mu = mean[(size_t)g * C + c], rs = rstd[(size_t)g * C + c], wc = w[c], bc = b[c];
    float a = 0.f, s = 0.f;
    for (long long r = r0; r < r1; ++r) {
      const float xh = (x[base + r * C + c] - mu) * rs;
      float dz = dy[base + r * C + c];
      if (gelu) dz *= gelu_d(xh * wc + bc);
      a += dz;
      s += dz * xh;
    }
    out[c] = a;
    out[C + c] = s;
  }
}

// dx = w rstd (dz - S_dz / n - xhat S_dzx / n), sums [G][2][C] over ALL ranks' rows, *count their number
__global__ __launch_bounds__(256) void cct_bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                const float* __restrict__ w, const float* __restrict__ b,
                                                                const float* __restrict__ sums, const float* __restrict__ count,
                                                                float* __restrict__ dx, long long R, int C, int G, int gelu) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)G * R * C) return;
  const int c = (int)(i % C);
  const int g = (int)(i / ((size_t)R * C));
  const float rs = rstd[(size_t)g * C + c], xh = (x[i] - mean[(size_t)g * C + c]) * rs;
  float dz = dy[i];
  if (gelu) dz *= gelu_d(xh * w[c] + b[c]);
  const float inv = 1.f / *count;
  dx[i] = w[c] * rs * (dz - sums[((size_t)g * 2) * C + c] * inv - xh * sums[((size_t)g * 2 + 1) * C + c] * inv);
}

// parameter gradients of a BatchNorm shared by the G groups: dw[c] = sum_g S_dzx (LOCAL sums: ranks are averaged by the
// data-parallel wrapper like every other parameter), db[c] = sum_g S_dz
__global__ __launch_bounds__(256) void cct_bn_param_grads_kernel(const float* __restrict__ sums, float* __restrict__ dw, float* __restrict__ db,
                                                                  int C, int G) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float a = 0.f, s = 0.f;
  for (int g = 0; g < G; ++g) {
    a += sums[((size_t)g * 2) * C + c];
    s += sums[((size_t)g * 2 + 1) * C + c];
  }
  db[c] = a;
  dw[c] = s;
}

// ---- one-channel BatchNorm over the mask logits: G groups of E contiguous elements ---------------------------------------------
// part[(g nblk + blk)][2]: with dy == nullptr: sum (x - s), sum (x - s)^2; else sum dy, sum dy xhat (xhat from mean[g], rstd[g])
__global__ __launch_bounds__(256) void cct_scalar_stats_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                const float* __restrict__ shift, const float* __restrict__ mean,
                                                                const float* __restrict__ rstd, float* __restrict__ part, size_t E,
                                                                size_t per_blk) {
  __shared__ float red[2][256];
  const int g = blockIdx.y, nblk = gridDim.x;
  const size_t e0 = (size_t)blockIdx.x * per_blk, e1 = e0 + per_blk < E ? e0 + per_blk : E;     // per_blk, E: multiples of 4
  const float4* xg = reinterpret_cast<const float4*>(x + (size_t)g * E);
  float a = 0.f, b = 0.f;
  if (!dy) {
    const float s = shift ? shift[0] : 0.f;
    for (size_t e = e0 / 4 + threadIdx.x; e < e1 / 4; e += 256) {
      const float4 v = xg[e];
      const float d0 = v.x - s, d1 = v.y - s, d2 = v.z - s, d3 = v.w - s;
      a += (d0 + d1) + (d2 + d3);
      b += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
  } else {
    const float mu = mean[g], rs = rstd[g];
    const float4* dg = reinterpret_cast<const float4*>(dy + (size_t)g * E);
    for (size_t e = e0 / 4 + threadIdx.x; e < e1 / 4; e += 256) {
      const float4 d = dg[e], v = xg[e];
      a += (d.x + d.y) + (d.z + d.w);
      b += (d.x * (v.x - mu) + d.y * (v.y - mu) + d.z * (v.z - mu) + d.w * (v.w - mu)) * rs;
    }
  }
  red[0][threadIdx.x] = a;
  red[1][threadIdx.x] = b;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      red[0][threadIdx.x] += red[0][threadIdx.x + o];
      red[1][threadIdx.x] += red[1][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    part[((size_t)g * nblk + blockIdx.x) * 2] = red[0][0];
    part[((size_t)g * nblk + blockIdx.x) * 2 + 1] = red[1][0];
  }
}

// out = (x - mean_g) rstd_g w + b
__global__ __launch_bounds__(256) void cct_scalar_bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                                   const float* __restrict__ rstd, const float* __restrict__ w,
                                                                   const float* __restrict__ b, float* __restrict__ out, size_t E4) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int g = blockIdx.y;
  if (i >= E4) return;
  const float mu = mean[g], sc = rstd[g] * w[0], bb = b[0];
  const float4 v = reinterpret_cast<const float4*>(x)[(size_t)g * E4 + i];
  reinterpret_cast<float4*>(out)[(size_t)g * E4 + i] = make_float4((v.x - mu) * sc + bb, (v.y - mu) * sc + bb, (v.z - mu) * sc + bb, (v.w - mu) * sc + bb);
}

// dx = w rstd (dy - S_dy / n - xhat S_dyx / n); sums [G][2]
__global__ __launch_bounds__(256) void cct_scalar_bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                       const float* __restrict__ w, const float* __restrict__ sums,
                                                                       const float* __restrict__ count, float* __restrict__ dx, size_t E4) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int g = blockIdx.y;
  if (i >= E4) return;
  const float mu = mean[g], rs = rstd[g], inv = 1.f / *count;
  const float k = w[0] * rs, a = sums[g * 2] * inv, s = sums[g * 2 + 1] * inv * rs;
  const float4 d = reinterpret_cast<const float4*>(dy)[(size_t)g * E4 + i], v = reinterpret_cast<const float4*>(x)[(size_t)g * E4 + i];
  reinterpret_cast<float4*>(dx)[(size_t)g * E4 + i] =
      make_float4(k * (d.x - a - (v.x - mu) * s), k * (d.y - a - (v.y - mu) * s), k * (d.z - a - (v.z - mu) * s), k * (d.w - a - (v.w - mu) * s));
}

// the same input gradient as three coefficients per group, dx = c1 dy + c2 x + c3, for the GEMM that consumes it (GemmLd::aff)
__global__ void cct_scalar_bn_bwd_coef_kernel(const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ w,
                                              const float* __restrict__ sums, const float* __restrict__ count, float* __restrict__ coef, int G) {
  const int g = threadIdx.x;
  if (g >= G) return;
  const float mu = mean[g], rs = rstd[g], inv = 1.f / *count;
  const float k = w[0] * rs, a = sums[g * 2] * inv, s = sums[g * 2 + 1] * inv * rs;
  coef[g * 3] = k;
  coef[g * 3 + 1] = -k * s;
  coef[g * 3 + 2] = k * (mu * s - a);
}

// ---- class-activation pooling (CC:48-50): per layer g and query q, softmax over the B*Tc entries (b, t) of a = CE . wa + ba,
//      pooled[g][q][c] = sum_e p_e CE[row_e][c].  Rows: m = (b Q + q) Tc + t.  One block per (q, g); C <= 1024, B*Tc <= 1024.
__global__ __launch_bounds__(256) void cct_act_pool_fwd_kernel(const float* __restrict__ ce, const float* __restrict__ wa, const float* __restrict__ ba,
                                                                float* __restrict__ p_out, float* __restrict__ pooled, int B, int Q, int Tc,
                                                                int C) {
  __shared__ float sa[1024];
  const int q = blockIdx.x, g = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ne = B * Tc;
  const size_t M = (size_t)B * Q * Tc;
  const float* ceg = ce + (size_t)g * M * C;
  for (int e = wave; e < ne; e += 4) {
    const size_t row = ((size_t)(e / Tc) * Q + q) * Tc + e % Tc;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += ceg[row * C + c] * wa[c];
    s = wave_total(s);
    if (lane == 0) sa[e] = s + ba[0];
  }
  __syncthreads();
  float mx = -INFINITY;
  for (int e = 0; e < ne; ++e) mx = fmaxf(mx, sa[e]);
  float den = 0.f;
  for (int e = 0; e < ne; ++e) den += expf(sa[e] - mx);
  const float inv = 1.f / den;
  for (int c = threadIdx.x; c < C; c += 256) {
    float acc = 0.f;
    for (int e = 0; e < ne; ++e) {
      const size_t row = ((size_t)(e / Tc) * Q + q) * Tc + e % Tc;
      acc += expf(sa[e] - mx) * inv * ceg[row * C + c];
    }
    pooled[((size_t)g * Q + q) * C + c] = acc;
  }
  for (int e = threadIdx.x; e < ne; e += 256) {
    const size_t row = ((size_t)(e / Tc) * Q + q) * Tc + e % Tc;
    p_out[(size_t)g * M + row] = expf(sa[e] - mx) * inv;
  }
}

// backward: dce[row_e][c] = p_e dpool[c] + da_e wa[c], da_e = p_e (dp_e - sum_e' p_e' dp_e'), dp_e = dpool . CE[row_e];
// part_wa[(g Q + q)][c] = sum_e da_e CE[row_e][c], part_ba[(g Q + q)] = sum_e da_e
__global__ __launch_bounds__(256) void cct_act_pool_bwd_kernel(const float* __restrict__ ce, const float* __restrict__ wa, const float* __restrict__ p,
                                                                const float* __restrict__ dpool, float* __restrict__ dce,
                                                                float* __restrict__ part_wa, float* __restrict__ part_ba, int B, int Q, int Tc,
                                                                int C) {
  __shared__ float sd[1024];
  const int q = blockIdx.x, g = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ne = B * Tc;
  const size_t M = (size_t)B * Q * Tc;
  const float* ceg = ce + (size_t)g * M * C;
  const float* pg = p + (size_t)g * M;
  const float* dpl = dpool + ((size_t)g * Q + q) * C;
  for (int e = wave; e < ne; e += 4) {
    const size_t row = ((size_t)(e / Tc) * Q + q) * Tc + e % Tc;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += ceg[row * C + c] * dpl[c];
    s = wave_total(s);
    if (lane == 0) sd[e] = s;
  }
  __syncthreads();
  float sdp = 0.f;
  for (int e = 0; e < ne; ++e) {
    const size_t row = ((size_t)(e / Tc) * Q + q) * Tc + e % Tc;
    sdp += pg[row] * sd[e];
  }
  float sba = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) {
    float acc = 0.f;
    for (int e = 0; e < ne; ++e) {
      const size_t row = ((size_t)(e / Tc) * Q + q) * Tc + e % Tc;
      const float pe = pg[row], da = pe * (sd[e] - sdp), v = ceg[row * C + c];
      dce[(size_t)g * M * C + row * C + c] = pe * dpl[c] + da * wa[c];
      acc += da * v;
      if (c == 0) sba += da;
    }
    part_wa[((size_t)g * Q + q) * C + c] = acc;
    if (c == 0) part_ba[(size_t)g * Q + q] = sba;
  }
}

// ---- small Linear with any output width (the class head: K1 = classes + 1 is not a multiple of 4): one wave per output -------
// y[r][k] = x[r] . w[k] + b[k] + (k == K1 - 1 ? last_bias : 0)        (add_bias_towards_void, CC:52)
__global__ __launch_bounds__(256) void cct_small_linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                                    float* __restrict__ y, int R, int C, int K1, float last_bias) {
  const long long o = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (o >= (long long)R * K1) return;
  const int r = (int)(o / K1), k = (int)(o - (long long)r * K1);
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += x[(size_t)r * C + c] * w[(size_t)k * C + c];
  s = wave_total(s);
  if (lane == 0) y[o] = s + b[k] + (k == K1 - 1 ? last_bias : 0.f);
}
// dx[r][c] = sum_k dy[r][k] w[k][c]
__global__ __launch_bounds__(256) void cct_small_linear_bwd_x_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx,
                                                                      int R, int C, int K1) {
  const int r = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += 256) {
    float s = 0.f;
#pragma unroll 8
    for (int k = 0; k < K1; ++k) s += dy[(size_t)r * K1 + k] * w[(size_t)k * C + c];
    dx[(size_t)r * C + c] = s;
  }
}
// partial[(split K1 + k)][C + 1]: sum over the split's rows r of dy[r][k] x[r][c] (columns 0 .. C-1) and of dy[r][k] (column C);
// grid (K1, splits); the caller adds the splits in order (cct_small_linear_bwd_w_final_kernel)
__global__ __launch_bounds__(256) void cct_small_linear_bwd_w_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ part,
                                                                      int R, int C, int K1) {
  const int k = blockIdx.x, sp = blockIdx.y, nsp = gridDim.y;
  const int rper = (R + nsp - 1) / nsp, r0 = sp * rper, r1 = min(R, r0 + rper);
  float* out = part + ((size_t)sp * K1 + k) * (C + 1);
  for (int c = threadIdx.x; c < C; c += 256) {
    float s = 0.f, sb = 0.f;
    for (int r = r0; r < r1; ++r) {
      const float d = dy[(size_t)r * K1 + k];
      s += d * x[(size_t)r * C + c];
      sb += d;
    }
    out[c] = s;
    if (c == 0) out[C] = sb;
  }
}
__global__ __launch_bounds__(256) void cct_small_linear_bwd_w_final_kernel(const float* __restrict__ part, float* __restrict__ dw, float* __restrict__ db,
                                                                            int nsp, int C, int K1) {
  const int k = blockIdx.x;
  for (int c = threadIdx.x; c <= C; c += 256) {
    float s = 0.f;
    for (int sp = 0; sp < nsp; ++sp) s += part[((size_t)sp * K1 + k) * (C + 1) + c];
    if (c < C) dw[(size_t)k * C + c] = s;
    else db[k] = s;
  }
}

// ---- layouts around the mask einsum ---------------------------------------------------------------------------------------
// kt[((b Tc + t) Cm + c) (G Q) + g Q + q] = mk[g][(b Q + q) Tc + t][c]: the per-clip mask kernels of all layers, contraction (c) major
__global__ __launch_bounds__(256) void cct_kern_pack_kernel(const float* __restrict__ mk, float* __restrict__ kt, int G, int B, int Q, int Tc, int Cm) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t n = (size_t)G * B * Q * Tc * Cm;
  if (i >= n) return;
  const int GQ = G * Q;
  const int gq = (int)(i % GQ);
  const size_t r = i / GQ;
  const int c = (int)(r % Cm);
  const int bt = (int)(r / Cm), b = bt / Tc, t = bt - b * Tc, g = gq / Q, q = gq - g * Q;
  kt[i] = mk[((size_t)g * B * Q * Tc + ((size_t)b * Q + q) * Tc + t) * Cm + c];
}
// dmk[g][(b Q + q) Tc + t][c] = dk[(b Tc + t)][g Q + q][c]
__global__ __launch_bounds__(256) void cct_kern_unpack_kernel(const float* __restrict__ dk, float* __restrict__ dmk, int G, int B, int Q, int Tc, int Cm) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t n = (size_t)G * B * Q * Tc * Cm;
  if (i >= n) return;
  const int c = (int)(i % Cm);
  const size_t r = i / Cm;                                  // (g, b, q, t)
  const int t = (int)(r % Tc);
  const size_t r2 = r / Tc;
  const int q = (int)(r2 % Q);
  const size_t r3 = r2 / Q;
  const int b = (int)(r3 % B), g = (int)(r3 / B);
  dmk[i] = dk[(((size_t)b * Tc + t) * (G * Q) + (size_t)g * Q + q) * Cm + c];
}

}  // namespace tr
}  // namespace axvs
