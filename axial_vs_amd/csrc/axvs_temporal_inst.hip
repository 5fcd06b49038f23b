// One translation unit per (operand type, frame count): the instantiations of temporal_fused_kernel are what dominates the
// build (5 key-step counts x {with, without FFN} large kernels each), so __graft_entry__.build() compiles them in parallel:
//   hipcc -c axvs_temporal_inst.hip -DAXVS_INST_BF=<0|1> -DAXVS_INST_T=<1..8>
#include "axvs_host.h"
#include "axvs_fused.h"

#if !defined(AXVS_INST_BF) || !defined(AXVS_INST_T)
#error "compile with -DAXVS_INST_BF=<0|1> -DAXVS_INST_T=<frames> [-DAXVS_INST_MT=<16-row tiles per workgroup>]"
#endif
#ifndef AXVS_INST_MT   // default tile: 64 rows (32 for T >= 5: the x tile takes T * 16 KiB of LDS per 32 rows); MT = 1 (16 rows) units serve problems with few rows (cross-clip queries)
#define AXVS_INST_MT (AXVS_INST_T <= 4 ? 4 : 2)
#endif

namespace axvs {

template <bool BF, int T, int MT, int NKS, bool FFN, int MQ = 0>
static int launch_one(unsigned grid, const TrajWs& w, const TrajPacked& p, const float* res, float* out, RowMap rm, long long Mp, int N, int L,
                      float scale, hipStream_t st, const FfnArgs* fa, int wt, const OwnQkv* oq = nullptr) {
  auto kern = &temporal_fused_kernel<BF, T, MT, NKS, FFN, MQ>;
  if (int rc = ensure_max_lds(reinterpret_cast<const void*>(kern))) return rc;
  const size_t lds = temporal_lds_bytes<T, MT, FFN, MQ>(FFN ? fa->F : 0);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, w.x16, p.wpq, p.bpq, p.wpkv, p.bpkv, p.wp, p.bp, res, out, rm, Mp, N, L, scale,
                     w.q16, w.k16, w.vt16, FFN ? *fa : FfnArgs{}, p.wk2t, (wt & 1) | (FFN ? wt & kOut16Mask : 0), (wt >> 1) & 3, FFN ? nullptr : p.post_ln_g,
                     FFN ? nullptr : p.post_ln_b, MQ ? *oq : OwnQkv{});
  return AXVS_OK;
}

template <bool BF, int T, int MT, int NKS>
static int launch_temporal_t(const TrajWs& w, const TrajPacked& p, const float* res, float* out, RowMap rm, long long Mp, int N, int L,
                             float scale, hipStream_t st, const FfnArgs* fa, int wt, const OwnQkv* oq) {
  // with the spatial half in the kernel every sequence gets its own ceil(N / rows) tiles (see temporal_fused_kernel)
  const unsigned grid = NKS > 0 ? (unsigned)((Mp / N) * ((N + MT * 16 - 1) / (MT * 16))) : (unsigned)((Mp + MT * 16 - 1) / (MT * 16));
  if constexpr (NKS > 0 && NKS <= 3 && MT == 4 && T <= 4) {
    if (oq) {                           // merged q/k/v + trajectory launch
      if constexpr (NKS == 2 && T >= 2) {
        if (L == 64) {                  // a 64-row tile is one frame: own frame first, from registers
          if (fa) return launch_one<BF, T, MT, NKS, true, 2>(grid, w, p, res, out, rm, Mp, N, L, scale, st, fa, wt, oq);
          return launch_one<BF, T, MT, NKS, false, 2>(grid, w, p, res, out, rm, Mp, N, L, scale, st, fa, wt, oq);
        }
      }
      if (fa) return launch_one<BF, T, MT, NKS, true, 1>(grid, w, p, res, out, rm, Mp, N, L, scale, st, fa, wt, oq);
      return launch_one<BF, T, MT, NKS, false, 1>(grid, w, p, res, out, rm, Mp, N, L, scale, st, fa, wt, oq);
    }
  }
  if constexpr (NKS > 0 && MT == 1 && T <= 4) {
    if (oq) {                           // 16-row tiles (few rows: pyramid levels of 32 x 32 and below, cross-clip queries)
      if (fa) return fail(AXVS_ERR_ARG, "internal: merged q/k/v on 16-row tiles excludes the FFN");
      return launch_one<BF, T, MT, NKS, false, 1>(grid, w, p, res, out, rm, Mp, N, L, scale, st, fa, wt, oq);
    }
  }
  if constexpr (NKS > 0 && NKS <= 3 && MT == 2 && T >= 5) {
    if (oq) {                           // 32-row tiles: 5 .. 8 frames per clip (Tube-Link's T = 5 levels)
      if (fa) return fail(AXVS_ERR_ARG, "internal: merged q/k/v on 32-row tiles excludes the FFN");
      return launch_one<BF, T, MT, NKS, false, 1>(grid, w, p, res, out, rm, Mp, N, L, scale, st, fa, wt, oq);
    }
  }
  if (oq) return fail(AXVS_ERR_ARG, "internal: merged q/k/v needs the in-kernel spatial half on 64- / 16-row tiles (T <= 4) or 32-row tiles (T = 5 .. 8)");
  if constexpr (NKS > 0 && MT == 4) {
    if (fa) return launch_one<BF, T, MT, NKS, true>(grid, w, p, res, out, rm, Mp, N, L, scale, st, fa, wt);   // trajectory attention + FFN
  }
  if (fa) return fail(AXVS_ERR_ARG, "internal: FFN fusion needs the in-kernel spatial half and 64-row tiles");
  return launch_one<BF, T, MT, NKS, false>(grid, w, p, res, out, rm, Mp, N, L, scale, st, fa, wt);
}

template <bool BF, int T, int MT>
int launch_temporal_n(int nks, const TrajWs& w, const TrajPacked& p, const float* res, float* out, RowMap rm, long long Mp, int N,
                      int L, float scale, hipStream_t st, const FfnArgs* fa, int wt, const OwnQkv* oq) {
  switch (nks) {
    case 0: return launch_temporal_t<BF, T, MT, 0>(w, p, res, out, rm, Mp, N, L, scale, st, fa, wt, oq);
    case 1: return launch_temporal_t<BF, T, MT, 1>(w, p, res, out, rm, Mp, N, L, scale, st, fa, wt, oq);
    case 2: return launch_temporal_t<BF, T, MT, 2>(w, p, res, out, rm, Mp, N, L, scale, st, fa, wt, oq);
    case 3: return launch_temporal_t<BF, T, MT, 3>(w, p, res, out, rm, Mp, N, L, scale, st, fa, wt, oq);
    case 4: return launch_temporal_t<BF, T, MT, 4>(w, p, res, out, rm, Mp, N, L, scale, st, fa, wt, oq);
    default: return fail(AXVS_ERR_ARG, "bad nks");
  }
}

template int launch_temporal_n<(AXVS_INST_BF != 0), AXVS_INST_T, AXVS_INST_MT>(
    int, const TrajWs&, const TrajPacked&, const float*, float*, RowMap, long long, int, int, float, hipStream_t, const FfnArgs*, int,
    const OwnQkv*);

}  // namespace axvs

#if defined(AXVS_STAMPS) && !defined(AXVS_STAMPS_QKV) && AXVS_INST_BF == 0 && AXVS_INST_T == 4 && AXVS_INST_MT == 4
// diagnostic builds: phase stamps of the f16 / T = 4 trajectory kernels (tools/stamps.py)
extern "C" int axvs_debug_read_stamps(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(axvs::g_stamps), sizeof(unsigned long long) * n);
}
#endif

#if defined(AXVS_STAMPS) && !defined(AXVS_STAMPS_QKV) && AXVS_INST_BF == 0 && AXVS_INST_T == 4 && AXVS_INST_MT == 1
// ... and of the 16-row kernels of the same (f16, T = 4): the cross-clip queries (tools/r5/cc_traj_stamps.py)
extern "C" int axvs_debug_read_stamps_mt1(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(axvs::g_stamps), sizeof(unsigned long long) * n);
}
#endif

#if defined(AXVS_STAMPS_WG) && AXVS_INST_BF == 0 && AXVS_INST_T == 4 && AXVS_INST_MT == 4
// diagnostic builds: start / end of every workgroup of the last f16 / T = 4 launches (tools/r5/wg_times.py)
extern "C" int axvs_debug_read_wg_times(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(axvs::g_wg), sizeof(unsigned long long) * n);
}
#endif
