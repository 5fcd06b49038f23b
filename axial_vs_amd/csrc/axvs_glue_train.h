// Training tier of the pixel decoder's glue (round 6; SURVEY 8f-4 / VERDICT r5 "missing" 4): the 1x1 convolution + GroupNorm projections
// (WC/msdeformattn.py:349-375, used at :412 and :434) forward AND backward, so that the within-clip module's train() mode runs no torch kernel between the
// backbone maps and its output maps.  fp32 tensors, the GEMMs on the training tier's split-precision kernels (axvs_train_gemm.h: three bf16 pieces
// forward -- fp32 accuracy --, two backward), GroupNorm statistics and their gradients in fp32 with fixed summation orders (no atomics).
//   forward   y = x W^T + b            [M = N HW rows, Cout]                   (token rows; an NCHW input is transposed once)
//             mean, rstd per (sample n, group g) over HW x Cout/G values
//             out = (y - mean) rstd gamma + beta
//   backward  xhat = (y - mean) rstd;   A[n,c] = sum_p d_out,  B[n,c] = sum_p d_out xhat
//             d_beta = sum_n A,  d_gamma = sum_n B,  S1[n,g] = sum_{c in g} gamma_c A[n,c],  S2[n,g] = sum_{c in g} gamma_c B[n,c]
//             d_y = rstd (gamma d_out - (S1 + xhat S2) / cnt)
//             d_W = d_y^T x,  d_b = column sums of d_y,  d_x = d_y W
// Included by axvs_train.hip only, INSIDE its `namespace axvs { namespace {` (like axvs_cc_train_host.h): the kernels have internal linkage.
#pragma once

// t [N][HW][C] token rows -> x [N][C][HW] (the inverse of nchw_to_tokens_kernel); 64 x 64 tiles through LDS, any HW, C a multiple of 4
__global__ __launch_bounds__(256) void gt_tokens_to_nchw_kernel(const float* __restrict__ t, float* __restrict__ x, int C, int HW, long long t_batch_stride, long long t_ld) {
  __shared__ float tile[64][65];
  const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64, n = blockIdx.z;
  const int rc = (threadIdx.x & 15) * 4, rp = threadIdx.x >> 4;      // read: 16 threads x float4 cover 64 channels of a pixel row
  const float* src = t + (long long)n * t_batch_stride + (long long)p0 * t_ld + c0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = rp + 16 * i;
    float4 v = {0.f, 0.f, 0.f, 0.f};
    if (p0 + p < HW && c0 + rc < C) v = *reinterpret_cast<const float4*>(src + (long long)p * t_ld + rc);
    tile[p][rc] = v.x; tile[p][rc + 1] = v.y; tile[p][rc + 2] = v.z; tile[p][rc + 3] = v.w;
  }
  __syncthreads();
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;            // write: tx = pixel, 4 channel rows at a time
  float* dst = x + ((long long)n * C + c0) * HW + p0;
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const int c = ty + 4 * i;
    if (c0 + c < C && p0 + tx < HW) dst[(long long)c * HW + tx] = tile[tx][c];
  }
}

// x [N][C][HW] -> token rows t [N][HW][C] (contiguous)
__global__ __launch_bounds__(256) void gt_nchw_to_tokens_kernel(const float* __restrict__ x, float* __restrict__ t, int C, int HW) {
  __shared__ float tile[64][65];
  const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64, n = blockIdx.z;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const float* src = x + ((long long)n * C + c0) * HW + p0;
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const int c = ty + 4 * i;
    tile[c][tx] = (c0 + c < C && p0 + tx < HW) ? src[(long long)c * HW + tx] : 0.f;
  }
  __syncthreads();
  const int wc = (threadIdx.x & 15) * 4, wp = threadIdx.x >> 4;
  float* dst = t + ((long long)n * HW + p0) * C + c0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = wp + 16 * i;
    if (p0 + p < HW && c0 + wc < C)
      *reinterpret_cast<float4*>(dst + (long long)p * C + wc) = float4{tile[wc][p], tile[wc + 1][p], tile[wc + 2][p], tile[wc + 3][p]};
  }
}

// token rows that are a slice of a wider buffer -> contiguous [N][HW][C]
__global__ __launch_bounds__(256) void gt_gather_tokens_kernel(const float* __restrict__ src, float* __restrict__ dst, int HW, int C, long long batch_stride, long long ld,
                                                               long long total4) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const int c4 = (int)(i % (C / 4));
  const long long r = i / (C / 4);
  const long long n = r / HW, p = r - n * HW;
  *reinterpret_cast<float4*>(dst + r * C + c4 * 4) = *reinterpret_cast<const float4*>(src + n * batch_stride + p * ld + c4 * 4);
}

// Per-(sample, 64-row block, channel) column sums: a = sum v, b = sum v w with w = 1 (statistics: v = y, "b" = sum y^2 through w = y) or w = xhat (backward).
// MODE 0: (sum y, sum y^2);  MODE 1: (sum d, sum d xhat), xhat = (y - mean[n,g]) rstd[n,g].   part [N][nblk][C][2], fixed order inside the block.
template <int MODE>
__global__ __launch_bounds__(256) void gt_block_colsums_kernel(const float* __restrict__ v, const float* __restrict__ y, const float* __restrict__ stats /* [N][G][2] */,
                                                               float* __restrict__ part, int HW, int C, int G) {
  const int n = blockIdx.y, r0 = blockIdx.x * 64, rows = min(64, HW - r0);
  const int cg = C / G;
  for (int c4 = threadIdx.x; c4 < C / 4; c4 += 256) {
    float4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
    float mu[4] = {0.f, 0.f, 0.f, 0.f}, rs[4] = {1.f, 1.f, 1.f, 1.f};
    if (MODE == 1) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int g = (c4 * 4 + k) / cg;
        mu[k] = stats[((long long)n * G + g) * 2];
        rs[k] = stats[((long long)n * G + g) * 2 + 1];
      }
    }
    for (int r = 0; r < rows; ++r) {
      const long long o = ((long long)n * HW + r0 + r) * C + c4 * 4;
      const float4 d = *reinterpret_cast<const float4*>(v + o);
      a.x += d.x; a.y += d.y; a.z += d.z; a.w += d.w;
      if (MODE == 0) {
        b.x += d.x * d.x; b.y += d.y * d.y; b.z += d.z * d.z; b.w += d.w * d.w;
      } else {
        const float4 yy = *reinterpret_cast<const float4*>(y + o);
        b.x += d.x * ((yy.x - mu[0]) * rs[0]); b.y += d.y * ((yy.y - mu[1]) * rs[1]);
        b.z += d.z * ((yy.z - mu[2]) * rs[2]); b.w += d.w * ((yy.w - mu[3]) * rs[3]);
      }
    }
    float* o = part + (((long long)n * gridDim.x + blockIdx.x) * C + c4 * 4) * 2;
    o[0] = a.x; o[1] = b.x; o[2] = a.y; o[3] = b.y; o[4] = a.z; o[5] = b.z; o[6] = a.w; o[7] = b.w;
  }
}

// ab [N][C][2] = sum over the blocks (block order)
__global__ __launch_bounds__(256) void gt_sum_blocks_kernel(const float* __restrict__ part, float* __restrict__ ab, int nblk, int C, long long total /* N * C */) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const long long n = i / C;
  const int c = (int)(i - n * C);
  float a = 0.f, b = 0.f;
  for (int k = 0; k < nblk; ++k) {
    const float* p = part + (((long long)n * nblk + k) * C + c) * 2;
    a += p[0]; b += p[1];
  }
  ab[i * 2] = a; ab[i * 2 + 1] = b;
}

// forward statistics: stats [N][G] = (mean, rstd) from ab = (sum y, sum y^2) per (n, c), channels of a group in order
__global__ __launch_bounds__(256) void gt_group_stats_kernel(const float* __restrict__ ab, float* __restrict__ stats, int C, int G, float cnt, float eps, int total /* N * G */) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int n = i / G, g = i - n * G, cg = C / G;
  double s = 0.0, q = 0.0;       // (a few hundred channel sums per group: double keeps E[y^2] - mean^2 from cancelling)
  for (int c = g * cg; c < (g + 1) * cg; ++c) { s += ab[((long long)n * C + c) * 2]; q += ab[((long long)n * C + c) * 2 + 1]; }
  const double mu = s / cnt, var = q / cnt - mu * mu;
  stats[i * 2] = (float)mu;
  stats[i * 2 + 1] = (float)(1.0 / sqrt((var > 0.0 ? var : 0.0) + (double)eps));
}

// out = (y - mean) rstd gamma + beta on token rows (contiguous y; out rows at n * out_batch_stride + p * out_ld)
__global__ __launch_bounds__(256) void gt_gn_apply_kernel(const float* __restrict__ y, const float* __restrict__ stats, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ out, int HW, int C, int G, long long out_batch_stride,
                                                          long long out_ld, long long total4) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const int c = (int)(i % (C / 4)) * 4;
  const long long r = i / (C / 4);
  const long long n = r / HW, p = r - n * HW;
  const float4 v = *reinterpret_cast<const float4*>(y + r * C + c);
  const float4 ga = *reinterpret_cast<const float4*>(gamma + c), be = *reinterpret_cast<const float4*>(beta + c);
  const int cg = C / G;
  float o[4] = {v.x, v.y, v.z, v.w};
  const float gg[4] = {ga.x, ga.y, ga.z, ga.w}, bb[4] = {be.x, be.y, be.z, be.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float* st = stats + (n * G + (c + k) / cg) * 2;
    o[k] = (o[k] - st[0]) * st[1] * gg[k] + bb[k];
  }
  *reinterpret_cast<float4*>(out + n * out_batch_stride + p * out_ld + c) = float4{o[0], o[1], o[2], o[3]};
}

// backward finalisation: d_beta[c] = sum_n A, d_gamma[c] = sum_n B (sample order); S [N][G][2] = sum_{c in g} gamma_c (A, B)
__global__ __launch_bounds__(256) void gt_gn_bwd_params_kernel(const float* __restrict__ ab, float* __restrict__ d_gamma, float* __restrict__ d_beta, int N, int C) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float a = 0.f, b = 0.f;
  for (int n = 0; n < N; ++n) { a += ab[((long long)n * C + c) * 2]; b += ab[((long long)n * C + c) * 2 + 1]; }
  d_beta[c] = a;
  d_gamma[c] = b;
}
__global__ __launch_bounds__(256) void gt_gn_bwd_groups_kernel(const float* __restrict__ ab, const float* __restrict__ gamma, float* __restrict__ S, int C, int G, int total) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int n = i / G, g = i - n * G, cg = C / G;
  float s1 = 0.f, s2 = 0.f;
  for (int c = g * cg; c < (g + 1) * cg; ++c) {
    s1 += gamma[c] * ab[((long long)n * C + c) * 2];
    s2 += gamma[c] * ab[((long long)n * C + c) * 2 + 1];
  }
  S[i * 2] = s1; S[i * 2 + 1] = s2;
}
// d_y = rstd (gamma d - (S1 + xhat S2) / cnt), written over d (contiguous token rows)
__global__ __launch_bounds__(256) void gt_gn_bwd_apply_kernel(float* __restrict__ d, const float* __restrict__ y, const float* __restrict__ stats, const float* __restrict__ S,
                                                              const float* __restrict__ gamma, int HW, int C, int G, float inv_cnt, long long total4) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const int c = (int)(i % (C / 4)) * 4;
  const long long r = i / (C / 4);
  const long long n = r / HW;
  const float4 dv = *reinterpret_cast<const float4*>(d + r * C + c), yv = *reinterpret_cast<const float4*>(y + r * C + c);
  const float4 ga = *reinterpret_cast<const float4*>(gamma + c);
  const int cg = C / G;
  float dd[4] = {dv.x, dv.y, dv.z, dv.w};
  const float yy[4] = {yv.x, yv.y, yv.z, yv.w}, gg[4] = {ga.x, ga.y, ga.z, ga.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long long sg = (n * G + (c + k) / cg) * 2;
    const float mu = stats[sg], rs = stats[sg + 1];
    const float xh = (yy[k] - mu) * rs;
    dd[k] = rs * (gg[k] * dd[k] - (S[sg] + xh * S[sg + 1]) * inv_cnt);
  }
  *reinterpret_cast<float4*>(d + r * C + c) = float4{dd[0], dd[1], dd[2], dd[3]};
}

