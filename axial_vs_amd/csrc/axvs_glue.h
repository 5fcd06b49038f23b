// Pixel-decoder glue around the within-clip stages: 1x1 conv + GroupNorm projections between backbone NCHW maps and
// token rows, 2-D sine position embedding.  Reference: WC/msdeformattn.py:349-375 (input_proj / output_proj), :404-435
// (forward_features), WC/pos_embeddings.py:12-53 (PositionEmbeddingSine).
#pragma once
#include "axvs_common.h"

namespace axvs {

// A-operand loader for a 1x1 convolution that reads the backbone map where it lies: x [N][Cin][HW] fp32 (NCHW), row m = n*HW + p,
// k = input channel.  Lanes of a wave hold consecutive m, so each of the 8 strided loads is coalesced across the wave.
// Split precision (hi | hi | lo along a 3x longer K, see ALoadRowsF32Split3): the projection output feeds a GroupNorm directly.
template <bool BF>
struct ALoadNCHWSplit3 {
  static constexpr int kPrefetch = 1;
  const float* x;
  int M, K, HW;          // M = N*HW rows, K = Cin
  __device__ __forceinline__ u16x8 load(int m, int k) const {
    m = min(m, M - 1);
    const int part = k / K, kk = k - part * K;
    const int n = m / HW, p = m - n * HW;
    const float* s = x + ((long long)n * K + kk) * HW + p;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = s[(long long)i * HW];
    u16x8 hi = cvt8<BF>(v);
    if (part < 2) return hi;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] -= H16<BF>::to_f32(hi[i]);
    return cvt8<BF>(v);
  }
};

// x [N][C][HW] fp32 (NCHW) -> token rows t [N][HW][C]: the transposing stage in front of the 128 x 128 three-piece GEMM for the input projections of the
// large pyramid levels (round 5: ALoadNCHWSplit3 above reads every element three times, 64 bytes at a time -- 63 TFLOP/s at [32786 x 256 x 512]).
// 64 channels x 64 pixels per workgroup through LDS; any HW (pixel rows start at 4-byte boundaries), C a multiple of 4.
__global__ __launch_bounds__(256) void nchw_to_tokens_kernel(const float* __restrict__ x, float* __restrict__ t, int C, int HW) {
  __shared__ float tile[64][65];
  const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64, n = blockIdx.z;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;      // read: tx = pixel, 4 channel rows at a time
  const float* src = x + ((long long)n * C + c0) * HW + p0;
  float v[16];                 // all 16 loads of a thread in flight before the first LDS store (round 6: four at a time left the kernel at 1.7 TB/s)
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = ty + 4 * i;
    v[i] = (c0 + c < C && p0 + tx < HW) ? src[(long long)c * HW + tx] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) tile[ty + 4 * i][tx] = v[i];
  __syncthreads();
  // write: 16 threads x float4 cover the 64 channels of a pixel row (256 contiguous bytes), 16 pixels per round
  const int wc = (threadIdx.x & 15) * 4, wp = threadIdx.x >> 4;
  float* dst = t + ((long long)n * HW + p0) * C + c0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = wp + 16 * i;
    if (p0 + p < HW && c0 + wc < C)
      *reinterpret_cast<float4*>(dst + (long long)p * C + wc) = float4{tile[wc][p], tile[wc + 1][p], tile[wc + 2][p], tile[wc + 3][p]};
  }
}

// y[i] = bias[col(i)] + p_0[i] + p_1[i] + ... (z order): the split-K partials of the 128 x 128 GEMM
__global__ __launch_bounds__(256) void splitk_sum_bias_kernel(const float* __restrict__ part, float* __restrict__ y, int Z, long long stride, const float* __restrict__ bias,
                                                              int N, long long tot4) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= tot4) return;
  float4 a = *reinterpret_cast<const float4*>(part + 4 * i);
  for (int z = 1; z < Z; ++z) {
    const float4 p = *reinterpret_cast<const float4*>(part + z * stride + 4 * i);
    a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w;
  }
  const float4 b = *reinterpret_cast<const float4*>(bias + (int)((4 * i) % N));
  *reinterpret_cast<float4*>(y + 4 * i) = float4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w};
}

// token rows that may be a slice of a wider buffer: row (n, p) at x + n*batch_stride + p*ld  (split precision as above)
template <bool BF>
struct ALoadTokensSplit3 {
  static constexpr int kPrefetch = 1;
  const float* x;
  int M, K, HW;
  long long batch_stride, ld;
  __device__ __forceinline__ u16x8 load(int m, int k) const {
    m = min(m, M - 1);
    const int part = k / K, kk = k - part * K;
    const int n = m / HW, p = m - n * HW;
    const float4* s = reinterpret_cast<const float4*>(x + n * batch_stride + p * ld + kk);
    const float4 a = s[0], b = s[1];
    float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    u16x8 hi = cvt8<BF>(v);
    if (part < 2) return hi;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] -= H16<BF>::to_f32(hi[i]);
    return cvt8<BF>(v);
  }
};

// GroupNorm statistics over token rows y [N*HW][C], deterministic (no atomics: a 1-ulp change of a statistic flips 16-bit
// roundings downstream and shows up as run-to-run differences of whole f16 ulps).  Stage 1: one workgroup = 64 rows of one
// sample, thread = 4 consecutive channels -> per-channel sums in LDS -> per-group partials [N][nblk][G][2] in channel order.
// Stage 2 (gn_finalize_kernel): partials summed in block order -> stats [N][G][2] = (sum, sum of squares).
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ y, float* __restrict__ partial, int HW, int C, int G) {
  extern __shared__ float chs[];                        // [RG][C] sums | [RG][C] sums of squares (RG row groups)
  const int n = blockIdx.y, r0 = blockIdx.x * 64, tid = threadIdx.x;
  const int rows = min(64, HW - r0);
  // threads = `lanes` float4 columns x RG row groups (C = 256: 64 x 4, 16 rows each, loads independent of one another): the
  // serial 64-row loop of round 1 made this kernel a 18 us latency chain whatever the problem size
  const int n4 = C / 4, lanes = n4 < 256 ? n4 : 256, RG = 256 / lanes;
  const int col = tid % lanes, g = tid / lanes;
  float* chq = chs + RG * C;
  for (int c0 = 0; c0 < n4; c0 += lanes) {
    const int c4 = c0 + col;
    float4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
    if (g < RG && c4 < n4) {
#pragma unroll 4
      for (int r = g; r < rows; r += RG) {
        const float4 v = *reinterpret_cast<const float4*>(y + ((long long)n * HW + r0 + r) * C + c4 * 4);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        q.x += v.x * v.x; q.y += v.y * v.y; q.z += v.z * v.z; q.w += v.w * v.w;
      }
      *reinterpret_cast<float4*>(chs + g * C + c4 * 4) = s;
      *reinterpret_cast<float4*>(chq + g * C + c4 * 4) = q;
    }
  }
  __syncthreads();
  const int cg = C / G;
  for (int gg = tid; gg < G; gg += 256) {
    float s = 0.f, q = 0.f;
    for (int j = 0; j < RG; ++j)                          // fixed order: row groups, then channels
      for (int c = gg * cg; c < (gg + 1) * cg; ++c) { s += chs[j * C + c]; q += chq[j * C + c]; }
    float* o = partial + (((long long)n * gridDim.x + blockIdx.x) * G + gg) * 2;
    o[0] = s; o[1] = q;
  }
}

// (Round 4: the second stage -- per (n, g): lanes stride over the blocks' partials, then a fixed-order butterfly -- is the first thing
//  gn_apply_kernel does for the groups its 64 channels touch, instead of a launch of its own: 3 launches per GroupNorm -> 2.)
// GroupNorm apply: out = (y - mean_g) * rstd_g * gamma_c + beta_c; token rows in, token rows (in place allowed) or NCHW out.
// Workgroup = 64 pixels x 64 channels of one sample, transposed through LDS for the NCHW store.
template <bool OUT_NCHW>
__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ y, const float* __restrict__ partial, int nblk,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float* __restrict__ out, int HW, int C, int G, float eps, long long out_ld,
                                                       long long out_batch_stride) {
  __shared__ float tile[64][65];
  __shared__ float gmu[64], grs[64];                   // mean / rstd of the (at most 33: channels per group >= 2) groups of my 64 channels
  const int n = blockIdx.z, p0 = blockIdx.x * 64, c0 = blockIdx.y * 64, tid = threadIdx.x;
  const int cg = C / G;
  const float cnt = (float)HW * cg;
  {
    // statistics of my groups from the blocks' partial sums, in a fixed order (every workgroup that needs a group computes the same
    // bits): one wave per group, lanes stride over the blocks, xor butterfly
    const int g_first = c0 / cg, g_last = min(c0 + 63, C - 1) / cg, lane = tid & 63;
    for (int g = g_first + (tid >> 6); g <= g_last; g += 4) {
      float s = 0.f, q = 0.f;
      // eight partials per lane in flight at a time, added in block order as before (round 6: one dependent load per 64 blocks made this prologue a
      // chain of L2 round trips in front of EVERY workgroup's 16 KB of work on the large maps: 257 blocks at the shipped VIPSeg res3 size)
      for (int b0 = lane; b0 < nblk; b0 += 64 * 8) {
        float2 pv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int b = b0 + 64 * u;
          pv[u] = b < nblk ? *reinterpret_cast<const float2*>(partial + (((long long)n * nblk + b) * G + g) * 2) : float2{0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { s += pv[u].x; q += pv[u].y; }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o, 64);
        q += __shfl_xor(q, o, 64);
      }
      if (lane == 0) {
        const float mu = s / cnt;
        gmu[g - g_first] = mu;
        grs[g - g_first] = rsqrtf(fmaxf(q / cnt - mu * mu, 0.f) + eps);
      }
    }
    __syncthreads();
  }
  const int g_first = c0 / cg;
  // read: thread -> (pixel row, 16 channels) as float4 x 4: 64 rows x 16 float4
  for (int i = tid; i < 64 * 16; i += 256) {
    const int r = i >> 4, c4 = (i & 15) * 4;
    const int p = p0 + r, c = c0 + c4;
    float4 v = {0.f, 0.f, 0.f, 0.f};
    if (p < HW && c < C) {
      v = *reinterpret_cast<const float4*>(y + ((long long)n * HW + p) * C + c);
      float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int g = (c + k) / cg - g_first;
        o[k] = (o[k] - gmu[g]) * grs[g] * gamma[c + k] + beta[c + k];
      }
      v = float4{o[0], o[1], o[2], o[3]};
      if (!OUT_NCHW) *reinterpret_cast<float4*>(out + (long long)n * out_batch_stride + (long long)p * out_ld + c) = v;
    }
    if (OUT_NCHW) {
      tile[r][c4] = v.x; tile[r][c4 + 1] = v.y; tile[r][c4 + 2] = v.z; tile[r][c4 + 3] = v.w;
    }
  }
  if (OUT_NCHW) {
    __syncthreads();
    for (int i = tid; i < 64 * 64; i += 256) {
      const int cc = i >> 6, r = i & 63;
      if (p0 + r < HW && c0 + cc < C) out[(long long)n * out_batch_stride + (long long)(c0 + cc) * HW + p0 + r] = tile[r][cc];
    }
  }
}

// PositionEmbeddingSine(num_pos_feats = C/2, normalize, mask = None) in token form, plus an optional per-channel vector (the
// level embedding, WC/msdeformattn.py:113-114): pos[n][row0 + y*W + x][c], rows of a level inside a [N][S][C] buffer.
__global__ void pos2d_kernel(float* __restrict__ pos, const float* __restrict__ add, int N, int H, int W, int C, long long S, long long row0,
                             float temperature, int normalize, float scale) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)H * W * C;
  if (idx >= total) return;
  const int c = (int)(idx % C);
  const long long r = idx / C;
  const int x = (int)(r % W), y = (int)(r / W);
  const int n = C / 2;
  float ye = (float)(y + 1), xe = (float)(x + 1);
  if (normalize) {
    const float eps = 1e-6f;
    ye = ye / ((float)H + eps) * scale;
    xe = xe / ((float)W + eps) * scale;
  }
  const int cc = c < n ? c : c - n;                       // first half: y, second half: x (pos_embeddings.py:52)
  const float dim_t = powf(temperature, 2.f * (float)(cc / 2) / (float)n);
  const float a = (c < n ? ye : xe) / dim_t;
  float v = (cc & 1) ? cosf(a) : sinf(a);
  if (add) v += add[c];
  for (int b = 0; b < N; ++b) pos[((long long)b * S + row0 + r) * C + c] = v;
}

// x[i] += v[i % C]  (level embedding added to a channels-last position embedding, WC/msdeformattn.py:117-118)
__global__ void add_channel_vector_kernel(float* __restrict__ x, const float* __restrict__ v, size_t n, int C) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] += v[i % C];
}

}  // namespace axvs
