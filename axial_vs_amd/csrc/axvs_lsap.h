// Clip-to-clip query alignment: cosine cost + linear sum assignment on the device, so the cross-clip module's
// input needs no host round trip.  Reference call site: MaXTron_Video-kMaX/maxtron_deeplab/maxtron_cc_model.py:360-369
// (`match_from_embds`: normalise, cos_sim = cur @ tgt^T, C = 1 - cos_sim, scipy.optimize.linear_sum_assignment(C^T)[1]);
// Tube-Link: models/video/tube_link_vis/mask2former_video_cc_head.py:907-913.
// The assignment algorithm lives in a third-party dependency of the reference that is not under /root/reference:
// SciPy (unpinned by the reference; 1.15.3 in this image), scipy/optimize/rectangular_lsap/rectangular_lsap.cpp --
// the shortest-augmenting-path algorithm of D. F. Crouse, "On implementing 2D rectangular assignment algorithms", IEEE
// Trans. Aerospace and Electronic Systems 52(4), 2016.  Restated here step for step (double-precision duals, the same
// scan order through the `remaining` list, the same tie rule) so that ties break identically.
#pragma once
#include "axvs_common.h"

namespace axvs {

// rows normalised like the reference (`x / x.norm(dim=1)[:, None]`): one wave per row of tgt (rows 0..Q-1) or cur (Q..2Q-1)
__global__ __launch_bounds__(256) void normalize_rows_kernel(const float* __restrict__ tgt, const float* __restrict__ cur,
                                                             float* __restrict__ out, int Q, int C) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= 2 * Q) return;
  const float* x = row < Q ? tgt + (long long)row * C : cur + (long long)(row - Q) * C;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += x[c] * x[c];
  const float nrm = sqrtf(wave_sum(s));
  for (int c = lane; c < C; c += 64) out[(long long)row * C + c] = x[c] / nrm;
}

// cost[t][j] = 1 - <tgt_t / |tgt_t|, cur_j / |cur_j|>   (rows = target queries, columns = current queries: C^T of the reference)
__global__ __launch_bounds__(256) void cosine_cost_kernel(const float* __restrict__ nrm, float* __restrict__ cost, int Q, int C) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= Q * Q) return;
  const int t = idx / Q, j = idx - t * Q;
  const float* a = nrm + (long long)t * C;
  const float* b = nrm + (long long)(Q + j) * C;
  float dot = 0.f;
  for (int c = 0; c < C; ++c) dot += b[c] * a[c];
  cost[idx] = 1.f - dot;
}

// candidate of the column scan: (shortest path cost, tie key).  The sequential rule
// `if (spc[j] < lowest || (spc[j] == lowest && row4col[j] == -1))` scanned in increasing position `it` picks the lowest value;
// among equals the LAST unassigned column if there is one, otherwise the FIRST column.  As a total order with "smaller wins":
// tie key = -1 - it for an unassigned column (always beats an assigned one, later position beats earlier), it for an assigned one.
struct LsapCand {
  double v;
  int k2;
};
__device__ __forceinline__ int lsap_key(bool unassigned, int it) { return unassigned ? -1 - it : it; }
__device__ __forceinline__ bool lsap_better(const LsapCand& a, const LsapCand& b) {   // a wins over b
  return a.v < b.v || (a.v == b.v && a.k2 < b.k2);
}
// wave-wide arg-min on the VALU (DPP within rows of 16, v_permlane*_swap across rows): no LDS crossbar round trips
template <int CTRL>
__device__ __forceinline__ LsapCand lsap_dpp(const LsapCand& c) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, c.v);
  const unsigned lo = __builtin_amdgcn_update_dpp(0u, (unsigned)b, CTRL, 0xF, 0xF, true);
  const unsigned hi = __builtin_amdgcn_update_dpp(0u, (unsigned)(b >> 32), CTRL, 0xF, 0xF, true);
  LsapCand o;
  o.v = __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
  o.k2 = (int)__builtin_amdgcn_update_dpp(0u, (unsigned)c.k2, CTRL, 0xF, 0xF, true);
  return o;
}
__device__ __forceinline__ LsapCand lsap_wave_best(LsapCand best) {
  LsapCand o = lsap_dpp<0xB1>(best);  if (lsap_better(o, best)) best = o;      // lane ^ 1
  o = lsap_dpp<0x4E>(best);           if (lsap_better(o, best)) best = o;      // lane ^ 2
  o = lsap_dpp<0x141>(best);          if (lsap_better(o, best)) best = o;      // row_half_mirror (stands in for ^ 4)
  o = lsap_dpp<0x140>(best);          if (lsap_better(o, best)) best = o;      // row_mirror      (stands in for ^ 8)
  const unsigned long long b = __builtin_bit_cast(unsigned long long, best.v);
  {
    auto rl = __builtin_amdgcn_permlane16_swap((unsigned)b, (unsigned)b, false, false);
    auto rh = __builtin_amdgcn_permlane16_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
    auto rk = __builtin_amdgcn_permlane16_swap((unsigned)best.k2, (unsigned)best.k2, false, false);
    LsapCand c0{__builtin_bit_cast(double, ((unsigned long long)rh[0] << 32) | rl[0]), (int)rk[0]};
    LsapCand c1{__builtin_bit_cast(double, ((unsigned long long)rh[1] << 32) | rl[1]), (int)rk[1]};
    best = lsap_better(c1, c0) ? c1 : c0;
  }
  const unsigned long long b2 = __builtin_bit_cast(unsigned long long, best.v);
  {
    auto rl = __builtin_amdgcn_permlane32_swap((unsigned)b2, (unsigned)b2, false, false);
    auto rh = __builtin_amdgcn_permlane32_swap((unsigned)(b2 >> 32), (unsigned)(b2 >> 32), false, false);
    auto rk = __builtin_amdgcn_permlane32_swap((unsigned)best.k2, (unsigned)best.k2, false, false);
    LsapCand c0{__builtin_bit_cast(double, ((unsigned long long)rh[0] << 32) | rl[0]), (int)rk[0]};
    LsapCand c1{__builtin_bit_cast(double, ((unsigned long long)rh[1] << 32) | rl[1]), (int)rk[1]};
    best = lsap_better(c1, c0) ? c1 : c0;
  }
  return best;
}

// One wave per assignment problem (blockIdx.x = problem), n x n costs (fp32 -> double), n <= kLsapMax.
// out[t] = column assigned to row t (int64, like the reference's indices[1]).
constexpr int kLsapMax = 512;
// LDS_COST: the cost matrix is copied into dynamic LDS first (n*n*4 bytes <= 128 KiB): every augmenting step reads one row of
// it, and an L2 round trip per step dominated the run time.
template <bool LDS_COST>
__global__ __launch_bounds__(64) void lsap_kernel(const float* __restrict__ cost_all, long long* __restrict__ out_all, int n) {
  extern __shared__ float scost[];
  __shared__ double u[kLsapMax], v[kLsapMax], spc[kLsapMax];
  __shared__ int path[kLsapMax], row4col[kLsapMax], col4row[kLsapMax], remaining[kLsapMax];
  __shared__ unsigned char SR[kLsapMax], SC[kLsapMax];
  const int lane = threadIdx.x;
  const float* gcost = cost_all + (long long)blockIdx.x * n * n;
  long long* out = out_all + (long long)blockIdx.x * n;
  if (LDS_COST) {
    for (int i = lane; i < n * n; i += 64) scost[i] = gcost[i];
  }
  const float* cost = LDS_COST ? scost : gcost;
  for (int i = lane; i < n; i += 64) { u[i] = 0.0; v[i] = 0.0; row4col[i] = -1; col4row[i] = -1; path[i] = -1; }
  __syncthreads();
  const double INF = __builtin_huge_val();
  for (int curRow = 0; curRow < n; ++curRow) {
    // ---- augmenting_path ----
    double minVal = 0.0;
    int num_remaining = n;
    for (int it = lane; it < n; it += 64) { remaining[it] = n - it - 1; SR[it] = 0; SC[it] = 0; spc[it] = INF; }
    __syncthreads();
    int sink = -1, i = curRow;
    while (sink == -1) {
      if (lane == 0) SR[i] = 1;
      const double ui = u[i];
      LsapCand best{INF, 0x7fffffff};
      for (int it = lane; it < num_remaining; it += 64) {
        const int j = remaining[it];
        const double r = minVal + (double)cost[(long long)i * n + j] - ui - v[j];
        if (r < spc[j]) { path[j] = i; spc[j] = r; }
        const LsapCand c{spc[j], lsap_key(row4col[j] == -1, it)};
        if (lsap_better(c, best)) best = c;
      }
      best = lsap_wave_best(best);
      minVal = best.v;
      if (minVal == INF) { sink = -2; break; }                 // infeasible (cannot happen with finite costs)
      const int index = best.k2 < 0 ? -1 - best.k2 : best.k2;
      const int j = remaining[index];
      __syncthreads();                                          // every lane has read remaining[index]
      if (row4col[j] == -1) sink = j; else i = row4col[j];
      if (lane == 0) { SC[j] = 1; remaining[index] = remaining[num_remaining - 1]; }
      --num_remaining;
      __syncthreads();
    }
    if (sink < 0) break;
    // ---- update the dual variables ----
    if (lane == 0) u[curRow] += minVal;
    __syncthreads();
    for (int r = lane; r < n; r += 64)
      if (SR[r] && r != curRow) u[r] += minVal - spc[col4row[r]];
    for (int c = lane; c < n; c += 64)
      if (SC[c]) v[c] -= minVal - spc[c];
    __syncthreads();
    // ---- augment the previous solution ----
    if (lane == 0) {
      int j = sink;
      while (true) {
        const int r = path[j];
        row4col[j] = r;
        const int t = col4row[r];
        col4row[r] = j;
        j = t;
        if (r == curRow) break;
      }
    }
    __syncthreads();
  }
  for (int r = lane; r < n; r += 64) out[r] = col4row[r];
}

}  // namespace axvs
