// Small kernels around the GEMMs: weight packing, LayerNorm, the temporal attention (v1, unfused),
// positional embedding, scaled residual.
#pragma once
#include "axvs_common.h"

namespace axvs {

// ---------------- weight packing: fp32 nn.Linear [Nout, K] -> blocked 16-bit weight layout (wblk_off, axvs_common.h) ----------------
// Either dimension may be "head structured": `parts` consecutive groups of (heads x d) channels, each head padded
// from d to 32 (zero rows / columns), so that kernels always see 32-wide head blocks.
// Stored position p (0..31) of a permuted head block holds channel perm32(p): the order in which an MFMA D tile pair
// (two 16-row tiles, rows 4g+r) leaves 8 consecutive values per lane, so attention/QKV epilogues store 16 B per lane.
__device__ __host__ inline int perm32(int p) { return ((p >> 2) & 1) * 16 + (p >> 3) * 4 + (p & 3); }

struct PackDim {
  int orig;    // original size
  int padded;  // padded size
  int heads;   // 0: plain (identity up to `orig`, zero beyond); >0: head structured
  int d;       // head dim (when heads > 0)
  int perm;    // head structured only: positions within a 32-block are stored in perm32 order
  __device__ __host__ int to_orig(int i) const {
    if (heads == 0) return i < orig ? i : -1;
    int per = heads * 32;
    int part = i / per, w = i - part * per;
    int h = w >> 5, dd = perm ? perm32(w & 31) : (w & 31);
    if (dd >= d) return -1;
    int o = part * heads * d + h * d + dd;
    return o < orig ? o : -1;
  }
};

// n_off / n_total: the rows written are rows [n_off, n_off + nd.padded) of a wider blocked matrix with n_total rows
// (several nn.Linear weights stacked along the output dimension).
template <bool BF>
__global__ void pack_weight_kernel(const float* __restrict__ W, u16* __restrict__ out, PackDim nd, PackDim kd, int n_off,
                                   int n_total) {
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long total = (long long)nd.padded * kd.padded;
  if (idx >= total) return;
  int kk = idx & 31;
  long long t = idx >> 5;
  int n = t % nd.padded;
  int kb = t / nd.padded;
  int no = nd.to_orig(n), ko = kd.to_orig(kb * 32 + kk);
  float v = (no >= 0 && ko >= 0) ? W[(long long)no * kd.orig + ko] : 0.f;
  out[wblk_off(n_total, n_off + n, kb * 32 + kk)] = H16<BF>::from_f32(v);
}

// split-precision weights for the (hi | hi | lo) activations of ALoad*Split3: out blocked [3*Kp/32][N][32], K parts (hi | lo | hi)
template <bool BF>
__global__ void pack_weight_split3_kernel(const float* __restrict__ W, u16* __restrict__ out, PackDim nd, PackDim kd, int n_off,
                                          int n_total) {
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long total = (long long)nd.padded * kd.padded;
  if (idx >= total) return;
  int kk = idx & 31;
  long long t = idx >> 5;
  int n = t % nd.padded;
  int kb = t / nd.padded;
  int no = nd.to_orig(n), ko = kd.to_orig(kb * 32 + kk);
  float v = (no >= 0 && ko >= 0) ? W[(long long)no * kd.orig + ko] : 0.f;
  const u16 hi = H16<BF>::from_f32(v);
  const u16 lo = H16<BF>::from_f32(v - H16<BF>::to_f32(hi));
  const long long part = (long long)(kd.padded / 32) * ((n_total + 15) & ~15) * 32;
  const long long o = wblk_off(n_total, n_off + n, kb * 32 + kk);
  out[o] = hi;
  out[part + o] = lo;
  out[2 * part + o] = hi;
}

// Transposed k-half of proj_kv for the reassociated temporal logits (axvs_fused.h):  qk_h = Wk2_h^T q2_h.
// W: proj_kv weight [2C][C] fp32 (rows 0..C-1 = the k half), C = heads*32.  out: blocked [1][heads*C][32]:
// row h*C + c (c = input channel), position p = Wk2[h*32 + perm32(p)][c]  (K = head dim in the order of an MFMA D tile pair).
template <bool BF>
__global__ void pack_wk2t_kernel(const float* __restrict__ W, u16* __restrict__ out, int C, int heads) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)heads * C * 32) return;
  const int p = idx & 31;
  const long long r = idx >> 5;
  const int c = (int)(r % C), h = (int)(r / C);
  out[wblk_off((long long)heads * C, r, p)] = H16<BF>::from_f32(W[(long long)(h * 32 + perm32(p)) * C + c]);
}

// Per-head weight copies for the REASSOCIATED temporal half of the shape-generic tier (any T; d = 32, C = heads * 32):
//   wk2n [heads][C positions n][32 d]:  Wk2[h*32 + d][channel of stored position n]  -- u_h = Wk2_h^T q2_h as a K = 32 GEMM per head
//   wv2h [heads][C/32 kb][32 r][32]:    Wv2[h*32 + r][channel of stored position kb*32 + kk] -- o_h = Wv2_h z_h as an N = 32 GEMM per head
// "stored position": x is stored with the 32 channels of every head block in perm32 order (see pack_traj).
template <bool BF>
__global__ void pack_wk2n_kernel(const float* __restrict__ W, u16* __restrict__ out, int C, int heads) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)heads * C * 32) return;
  const int d = idx & 31;
  const long long r = idx >> 5;
  const int n = (int)(r % C), h = (int)(r / C);
  const int ch = (n & ~31) + perm32(n & 31);
  out[(long long)h * C * 32 + wblk_off(C, n, d)] = H16<BF>::from_f32(W[(long long)(h * 32 + d) * C + ch]);
}
template <bool BF>
__global__ void pack_wv2h_kernel(const float* __restrict__ W, u16* __restrict__ out, int C, int heads) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)heads * C * 32) return;
  const int kk = idx & 31;
  long long t = idx >> 5;
  const int r = (int)(t & 31);
  t >>= 5;
  const int KB = C / 32, kb = (int)(t % KB), h = (int)(t / KB);
  out[(long long)h * C * 32 + wblk_off(32, r, kb * 32 + kk)] = H16<BF>::from_f32(W[(long long)(C + h * 32 + r) * C + kb * 32 + perm32(kk)]);
}

// Reassociated temporal attention over any number of frames (WC/temporal_attention.py:66-73 with proj_kv applied once instead
// of once per frame):  logit_f = u_h . x_f,  a = softmax_f,  z_h = sum_f a_f x_f   (u_h = Wk2_h^T q2_h; the k2 bias drops out of
// the softmax).  U, Z: fp32 [M][heads * C] (head-major); X16: blocked [C/32][T*Mp][32], row f*Mp + m.  One wave per token:
// lane = (head, 32-channel block) -- the 8 lanes of a head reduce with three xor-shuffles --, online softmax over the frames,
// x read once per frame (64 bytes per lane, the 8 heads share the lines).  C = 256, 8 heads.
template <bool BF>
__global__ __launch_bounds__(256) void temporal_stream_kernel(const float* __restrict__ U, const u16* __restrict__ X16, float* __restrict__ Z,
                                                              long long M, long long Mp, int T) {
  constexpr int C = 256, HC = 8 * C;
  const long long m = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const int lane = threadIdx.x & 63, h = lane >> 3, sub = lane & 7;
  float u[32], z[32];
  {
    const float* up = U + m * HC + h * C + sub * 32;
#pragma unroll
    for (int i = 0; i < 32; i += 4) {
      const float4 v = *reinterpret_cast<const float4*>(up + i);
      u[i] = v.x; u[i + 1] = v.y; u[i + 2] = v.z; u[i + 3] = v.w;
    }
  }
#pragma unroll
  for (int i = 0; i < 32; ++i) z[i] = 0.f;
  float mx = -INFINITY, sum = 0.f;
  const long long R = Mp * T;
  for (int f = 0; f < T; ++f) {
    const u16* xp = X16 + blk_off(R, (long long)f * Mp + m, sub * 32);
    u16x8 xv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) xv[i] = *reinterpret_cast<const u16x8*>(xp + i * 8);
    float x[32], p = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      x[i] = H16<BF>::to_f32(xv[i >> 3][i & 7]);
      p += u[i] * x[i];
    }
    p += __shfl_xor(p, 1, 64);
    p += __shfl_xor(p, 2, 64);
    p += __shfl_xor(p, 4, 64);
    const float nm = fmaxf(mx, p);
    const float corr = __expf(mx - nm), e = __expf(p - nm);
    sum = sum * corr + e;
#pragma unroll
    for (int i = 0; i < 32; ++i) z[i] = z[i] * corr + e * x[i];
    mx = nm;
  }
  const float inv = 1.f / sum;
  float* zp = Z + m * HC + h * C + sub * 32;
#pragma unroll
  for (int i = 0; i < 32; i += 4) *reinterpret_cast<float4*>(zp + i) = float4{z[i] * inv, z[i + 1] * inv, z[i + 2] * inv, z[i + 3] * inv};
}

__global__ void pack_bias_kernel(const float* __restrict__ b, float* __restrict__ out, PackDim nd) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nd.padded) return;
  int o = nd.to_orig(i);
  out[i] = o >= 0 ? b[o] : 0.f;
}

// ---------------- LayerNorm over C (one wave per row); optional blocked 16-bit copy for the next GEMM ----------------
template <bool BF>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ X, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ Y,
                                                        u16* __restrict__ Y16, long long M, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* x = X + row * C;
  float s = 0.f;
  for (int c = lane * 4; c < C; c += 256) {
    float4 v = *reinterpret_cast<const float4*>(x + c);
    s += v.x + v.y + v.z + v.w;
  }
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) s = wave_xor_sum(s, m);
  const float mu = s / C;
  float q = 0.f;
  for (int c = lane * 4; c < C; c += 256) {
    float4 v = *reinterpret_cast<const float4*>(x + c);
    float a = v.x - mu, b = v.y - mu, cc = v.z - mu, d = v.w - mu;
    q += a * a + b * b + cc * cc + d * d;
  }
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) q = wave_xor_sum(q, m);
  const float rstd = rsqrtf(q / C + eps);
  for (int c = lane * 4; c < C; c += 256) {
    float4 v = *reinterpret_cast<const float4*>(x + c);
    float4 g = *reinterpret_cast<const float4*>(gamma + c);
    float4 b = *reinterpret_cast<const float4*>(beta + c);
    f32x4 o = {(v.x - mu) * rstd * g.x + b.x, (v.y - mu) * rstd * g.y + b.y, (v.z - mu) * rstd * g.z + b.z,
               (v.w - mu) * rstd * g.w + b.w};
    if (Y) *reinterpret_cast<float4*>(Y + row * C + c) = float4{o[0], o[1], o[2], o[3]};
    if (Y16) *reinterpret_cast<u16x4*>(Y16 + blk_off(M, row, c)) = cvt4<BF>(o);
  }
}

// ---------------- temporal half, v1 (WC/temporal_attention.py:66-73) ----------------
// q2: fp32 [M, Cp] (already * scale);  kv2: fp32 [T*M, 2*Cp] (k2 | v2; row f*M + m);  O16: blocked [Cp/32][M][32].
// One thread per (token, head, 4 channels); 8 threads cooperate on a head's 32-channel dot product.
template <bool BF>
__global__ __launch_bounds__(256) void temporal_attn_kernel(const float* __restrict__ q2, const float* __restrict__ kv2,
                                                            u16* __restrict__ O16, long long M, int T, int heads) {
  const int Cp = heads * 32;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long grp = gid >> 3;                 // (token, head)
  const int sub = (int)(gid & 7);
  const long long m = min(grp / heads, M - 1);
  const bool valid = (grp / heads) < M;
  const int h = (int)(grp % heads);
  const int c = h * 32 + sub * 4;
  const float4 q = *reinterpret_cast<const float4*>(q2 + m * Cp + c);
  // online softmax over the frames: any T (whole-video cross-clip inference runs T = number of clips)
  float mx = -INFINITY, sum = 0.f;
  f32x4 o = {0.f, 0.f, 0.f, 0.f};
  for (int f = 0; f < T; ++f) {
    const float* row = kv2 + ((long long)f * M + m) * 2 * Cp + c;
    const float4 k = *reinterpret_cast<const float4*>(row);
    const float4 v = *reinterpret_cast<const float4*>(row + Cp);
    float p = q.x * k.x + q.y * k.y + q.z * k.z + q.w * k.w;
    p += __shfl_xor(p, 1, 64);
    p += __shfl_xor(p, 2, 64);
    p += __shfl_xor(p, 4, 64);
    const float nm = fmaxf(mx, p);
    const float corr = __expf(mx - nm), e = __expf(p - nm);
    sum = sum * corr + e;
    o[0] = o[0] * corr + e * v.x; o[1] = o[1] * corr + e * v.y; o[2] = o[2] * corr + e * v.z; o[3] = o[3] * corr + e * v.w;
    mx = nm;
  }
  o *= 1.f / sum;
  if (valid) *reinterpret_cast<u16x4*>(O16 + blk_off(M, m, c)) = cvt4<BF>(o);
}

// ---------------- PositionEmbeddingSine3D, mask=None, channels-last (WC/pos_embeddings.py:86-130) ----------------
__global__ void pos3d_kernel(float* __restrict__ pos, int B, int T, int H, int W, int C, float temperature, int normalize,
                             float scale) {
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long total = (long long)T * H * W * C;
  if (idx >= total) return;
  int c = idx % C;
  long long r = idx / C;
  int w = r % W; r /= W;
  int h = r % H;
  int t = r / H;
  const int n = C / 2;
  float z = (float)(t + 1), y = (float)(h + 1), x = (float)(w + 1);
  if (normalize) {
    const float eps = 1e-6f;
    z = z / ((float)T + eps) * scale;
    y = y / ((float)H + eps) * scale;
    x = x / ((float)W + eps) * scale;
  }
  int cc = c < n ? c : c - n;
  float dim_t = powf(temperature, 2.f * (float)(cc / 2) / (float)n);
  float a = (c < n ? y : x) / dim_t;
  float v = (cc & 1) ? cosf(a) : sinf(a);
  float dim_z = powf(temperature, 2.f * (float)(c / 2) / (float)C);
  float az = z / dim_z;
  v += (c & 1) ? cosf(az) : sinf(az);
  for (int b = 0; b < B; ++b) pos[(long long)b * total + idx] = v;
}

// The same embedding with a padding mask (WC/pos_embeddings.py:96-106): the coordinates are running counts of the unmasked
// positions along t / h / w (not_mask.cumsum), normalised by the count over the whole axis.  mask: uint8 [B,T,H,W], non-zero =
// padded.  One thread per (token, channel); the three short scans are re-done per channel (this is a set-up kernel).
__global__ void pos3d_masked_kernel(float* __restrict__ pos, const unsigned char* __restrict__ mask, int B, int T, int H, int W, int C,
                                    float temperature, int normalize, float scale) {
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long total = (long long)B * T * H * W * C;
  if (idx >= total) return;
  int c = idx % C;
  long long r = idx / C;
  int w = r % W; r /= W;
  int h = r % H; r /= H;
  int t = r % T;
  int b = r / T;
  const unsigned char* mb = mask + (long long)b * T * H * W;
  int zc = 0, zt = 0, yc = 0, yt = 0, xc = 0, xt = 0;
  for (int i = 0; i < T; ++i) {
    const int nm = mb[((long long)i * H + h) * W + w] == 0;
    zt += nm;
    if (i <= t) zc += nm;
  }
  for (int i = 0; i < H; ++i) {
    const int nm = mb[((long long)t * H + i) * W + w] == 0;
    yt += nm;
    if (i <= h) yc += nm;
  }
  for (int i = 0; i < W; ++i) {
    const int nm = mb[((long long)t * H + h) * W + i] == 0;
    xt += nm;
    if (i <= w) xc += nm;
  }
  float z = (float)zc, y = (float)yc, x = (float)xc;
  if (normalize) {
    const float eps = 1e-6f;
    z = z / ((float)zt + eps) * scale;
    y = y / ((float)yt + eps) * scale;
    x = x / ((float)xt + eps) * scale;
  }
  const int n = C / 2;
  int cc = c < n ? c : c - n;
  float dim_t = powf(temperature, 2.f * (float)(cc / 2) / (float)n);
  float a = (c < n ? y : x) / dim_t;
  float v = (cc & 1) ? cosf(a) : sinf(a);
  float dim_z = powf(temperature, 2.f * (float)(c / 2) / (float)C);
  float az = z / dim_z;
  v += (c & 1) ? cosf(az) : sinf(az);
  pos[idx] = v;
}

__global__ void scaled_residual_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                       const float* __restrict__ gamma, float* __restrict__ out, size_t n, int C) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a[i] + gamma[i % C] * b[i];
}

}  // namespace axvs
