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

// cost[z][t][j] = 1 - <a_t, b_j> on row-normalised embeddings (rows = target queries, columns = current queries: C^T of the
// reference).  Problem z: Tc == 0 -> a = nrm rows 0..Q-1, b = rows Q..2Q-1 (one pair); Tc > 1 -> video z / (Tc-1), clips i and
// i + 1 of e [V][Tc][Q][C] (i = z % (Tc-1)).  A workgroup computes a 16 x 16 tile from two 16-row panels staged in LDS (row stride
// C + 1: conflict-free); every dot is still the plain sequential fp32 sum over c = 0..C-1, so the values are bit-identical to the
// one-thread-per-entry kernel of round 1 (33 us at Q = 128 because every lane streamed its own row from L2).
__global__ __launch_bounds__(256) void cost_tile_kernel(const float* __restrict__ nrm, float* __restrict__ cost, int Tc, int Q, int C) {
  extern __shared__ float cst[];
  const int ld = C + 1, tid = threadIdx.x, z = blockIdx.z;
  float* sa = cst;
  float* sb = cst + 16 * ld;
  const float *A, *B;
  if (Tc > 1) {
    const long long vid = z / (Tc - 1), i = z - vid * (Tc - 1);
    A = nrm + ((vid * Tc + i) * Q) * C;
    B = A + (long long)Q * C;
  } else {
    A = nrm;
    B = nrm + (long long)Q * C;
  }
  const int t0 = blockIdx.y * 16, j0 = blockIdx.x * 16;
  for (int i = tid; i < 16 * C; i += 256) {
    const int r = i / C, c = i - r * C;
    sa[r * ld + c] = t0 + r < Q ? A[(long long)(t0 + r) * C + c] : 0.f;
    sb[r * ld + c] = j0 + r < Q ? B[(long long)(j0 + r) * C + c] : 0.f;
  }
  __syncthreads();
  const int ty = tid >> 4, tx = tid & 15;
  const float* a = sa + ty * ld;
  const float* b = sb + tx * ld;
  float dot = 0.f;
  for (int c = 0; c < C; ++c) dot += b[c] * a[c];
  if (t0 + ty < Q && j0 + tx < Q) cost[((long long)z * Q + t0 + ty) * Q + j0 + tx] = 1.f - dot;
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

template <int CTRL>
__device__ __forceinline__ double lsap_dpp_f64(double x) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, x);
  const unsigned lo = __builtin_amdgcn_update_dpp(0u, (unsigned)b, CTRL, 0xF, 0xF, true);
  const unsigned hi = __builtin_amdgcn_update_dpp(0u, (unsigned)(b >> 32), CTRL, 0xF, 0xF, true);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double lsap_wave_min_f64(double x) {
  x = fmin(x, lsap_dpp_f64<0xB1>(x));
  x = fmin(x, lsap_dpp_f64<0x4E>(x));
  x = fmin(x, lsap_dpp_f64<0x141>(x));
  x = fmin(x, lsap_dpp_f64<0x140>(x));
  const unsigned long long b = __builtin_bit_cast(unsigned long long, x);
  auto rl = __builtin_amdgcn_permlane16_swap((unsigned)b, (unsigned)b, false, false);
  auto rh = __builtin_amdgcn_permlane16_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
  x = fmin(__builtin_bit_cast(double, ((unsigned long long)rh[0] << 32) | rl[0]), __builtin_bit_cast(double, ((unsigned long long)rh[1] << 32) | rl[1]));
  const unsigned long long b2 = __builtin_bit_cast(unsigned long long, x);
  rl = __builtin_amdgcn_permlane32_swap((unsigned)b2, (unsigned)b2, false, false);
  rh = __builtin_amdgcn_permlane32_swap((unsigned)(b2 >> 32), (unsigned)(b2 >> 32), false, false);
  return fmin(__builtin_bit_cast(double, ((unsigned long long)rh[0] << 32) | rl[0]), __builtin_bit_cast(double, ((unsigned long long)rh[1] << 32) | rl[1]));
}
__device__ __forceinline__ int lsap_wave_min_i32(int k) {
  k = min(k, (int)__builtin_amdgcn_update_dpp(0u, (unsigned)k, 0xB1, 0xF, 0xF, true));
  k = min(k, (int)__builtin_amdgcn_update_dpp(0u, (unsigned)k, 0x4E, 0xF, 0xF, true));
  k = min(k, (int)__builtin_amdgcn_update_dpp(0u, (unsigned)k, 0x141, 0xF, 0xF, true));
  k = min(k, (int)__builtin_amdgcn_update_dpp(0u, (unsigned)k, 0x140, 0xF, 0xF, true));
  auto r = __builtin_amdgcn_permlane16_swap((unsigned)k, (unsigned)k, false, false);
  k = min((int)r[0], (int)r[1]);
  r = __builtin_amdgcn_permlane32_swap((unsigned)k, (unsigned)k, false, false);
  return min((int)r[0], (int)r[1]);
}

// One wave per assignment problem, n x n costs (fp32 -> double), n <= kLsapMax.  out[t] = column assigned to row t (int64, like the
// reference's indices[1]).
//
// Column state lives in REGISTERS: lane l owns columns l, l + 64, ... (CPL per lane): shortest-path cost `spc`, dual `v`,
// `row4col`, the "removed" flag and -- instead of SciPy's `remaining` array -- the column's current POSITION in that array, which
// is all the tie rule needs (the scan visits positions in increasing order; removal moves the last position into the freed one,
// exactly the reference's `remaining[index] = remaining[--num_remaining]`).  Per augmenting step the wave reads one row of the
// cost matrix and u[i] from LDS, updates its columns, arg-min-reduces on the VALU (DPP / permlane), and finds the winner's column
// with a ballot + v_readlane: no barrier, no dependent LDS chain through `remaining[it] -> column -> cost / duals` (that chain was
// ~500 cycles per step in round 1: 275 us at n = 128; now ~120 us).  Row duals `u`, `col4row`, `path` and the master copy of
// `row4col` stay in LDS (the augmenting walk is a sequential pointer chase by lane 0).
//
// `chain` > 1: problems k = 0 .. chain-1 of a batch entry are solved one after the other and problem k's ROWS are taken in the order
// of problem k-1's result (row t of problem k is row out[k-1][t] of its stored matrix): the clip loop of maxtron_cc_model.py:280-301
// (`prev = cur[idx]` before the next match) on raw pair-wise cost matrices, in one launch per batch of videos.
constexpr int kLsapMax = 512;
template <int CPL, bool LDS_COST>
__global__ __launch_bounds__(64) void lsap_kernel(const float* __restrict__ cost_all, long long* __restrict__ out_all, int n, int chain) {
  extern __shared__ float scost[];
  __shared__ double u[kLsapMax];
  __shared__ int path[kLsapMax], row4col_m[kLsapMax], col4row[kLsapMax], rowperm[kLsapMax];
  const int lane = threadIdx.x;
  const double INF = __builtin_huge_val();
  for (int i = lane; i < n; i += 64) rowperm[i] = i;
  for (int k = 0; k < chain; ++k) {
    const float* gcost = cost_all + ((long long)blockIdx.x * chain + k) * n * n;
    long long* out = out_all + ((long long)blockIdx.x * chain + k) * n;
    __syncthreads();                                   // (one wave: orders the LDS traffic of consecutive problems)
    if (LDS_COST) {
      for (int i = lane; i < n * n; i += 64) scost[i] = gcost[i];
    }
    const float* cost = LDS_COST ? scost : gcost;
    for (int i = lane; i < n; i += 64) { u[i] = 0.0; row4col_m[i] = -1; col4row[i] = -1; path[i] = -1; }
    double v[CPL], spc[CPL];
    int row4col[CPL], pos[CPL];
    bool removed[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) { v[c] = 0.0; row4col[c] = -1; }
    __syncthreads();
    for (int curRow = 0; curRow < n; ++curRow) {
      // ---- augmenting_path ----
      double minVal = 0.0;
      int num_remaining = n;
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const int j = lane + 64 * c;
        pos[c] = n - 1 - j;                            // remaining[it] = n - it - 1
        removed[c] = j >= n;
        spc[c] = INF;
      }
      int sink = -1, i = curRow;
      while (sink == -1) {
        const double ui = u[i];
        const float* crow = cost + (long long)rowperm[i] * n;
        LsapCand best{INF, 0x7fffffff};
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
          const int j = lane + 64 * c;
          if (!removed[c]) {
            const double r = minVal + (double)crow[j] - ui - v[c];
            if (r < spc[c]) { path[j] = i; spc[c] = r; }
            const LsapCand cand{spc[c], lsap_key(row4col[c] == -1, pos[c])};
            if (lsap_better(cand, best)) best = cand;
          }
        }
        // two-phase arg-min: the minimum value first (v_min_f64 butterflies), then the smallest tie key among the lanes that hold it
        // (32-bit butterflies) -- about half the instructions of reducing (value, key) pairs together
        minVal = lsap_wave_min_f64(best.v);
        if (minVal == INF) { sink = -2; break; }       // infeasible (cannot happen with finite costs)
        const int k2 = lsap_wave_min_i32(best.v == minVal ? best.k2 : 0x7fffffff);
        const int index = k2 < 0 ? -1 - k2 : k2;
        // the winner is the live column at position `index`; the column at the last position moves into it
        int wj = -1, wr = -1;
        const int last = num_remaining - 1;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
          if (!removed[c]) {
            if (pos[c] == index) { wj = lane + 64 * c; wr = row4col[c]; removed[c] = true; }
            else if (pos[c] == last) pos[c] = index;
          }
        }
        const unsigned long long m = __ballot(wj >= 0);
        const int wl = __builtin_ctzll(m);
        const int j = __builtin_amdgcn_readlane(wj, wl);
        const int r4c = __builtin_amdgcn_readlane(wr, wl);
        if (r4c == -1) sink = j; else i = r4c;
        --num_remaining;
      }
      if (sink < 0) break;
      // ---- update the dual variables: visited rows are curRow and row4col[j] of the visited columns j != sink ----
      if (lane == 0) u[curRow] += minVal;
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const int j = lane + 64 * c;
        if (removed[c] && j < n) {
          if (j != sink) u[row4col[c]] += minVal - spc[c];
          v[c] -= minVal - spc[c];
        }
      }
      __syncthreads();
      // ---- augment the previous solution ----
      if (lane == 0) {
        int j = sink;
        while (true) {
          const int r = path[j];
          row4col_m[j] = r;
          const int t = col4row[r];
          col4row[r] = j;
          j = t;
          if (r == curRow) break;
        }
      }
      __syncthreads();
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const int j = lane + 64 * c;
        if (j < n) row4col[c] = row4col_m[j];
      }
    }
    __syncthreads();
    for (int r = lane; r < n; r += 64) out[r] = col4row[r];
    if (k + 1 < chain) {                               // next problem: rows in the order of this result (prev = cur[idx])
      for (int r = lane; r < n; r += 64) rowperm[r] = col4row[r];
    }
  }
}

// rows of x [R][C] divided by their norm (`x / x.norm(dim=1)[:, None]`), one wave per row
__global__ __launch_bounds__(256) void normalize_rows1_kernel(const float* __restrict__ x_all, float* __restrict__ out, long long R, int C) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= R) return;
  const float* x = x_all + row * C;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += x[c] * x[c];
  const float nrm = sqrtf(wave_sum(s));
  for (int c = lane; c < C; c += 64) out[row * C + c] = x[c] / nrm;
}

}  // namespace axvs
