// Instantiations of the head-split trajectory kernel (axvs_hsplit.h), one translation unit per (operand type, group of three frame counts):
//   hipcc -c axvs_hsplit_inst.hip -DAXVS_HS_BF=<0|1> -DAXVS_HS_TG=<0..3>      frames 3 TG + 1 .. 3 TG + 3, 1 .. 4 key steps of 32 per frame
#include "axvs_host.h"
#include "axvs_hsplit.h"

#if !defined(AXVS_HS_BF) || !defined(AXVS_HS_TG)
#error "compile with -DAXVS_HS_BF=<0|1> -DAXVS_HS_TG=<0..3>"
#endif

namespace axvs {

template <bool BF, int T, int NKS>
static int hs_launch_one(unsigned grid, hipStream_t st, const HsArgs& a) {
  hipLaunchKernelGGL((hs_traj_kernel<BF, T, NKS>), dim3(grid), dim3(256), 0, st, a);
  return AXVS_OK;
}
template <bool BF, int T>
static int hs_launch_t(int nks, unsigned grid, hipStream_t st, const HsArgs& a) {
  switch (nks) {
    case 1: return hs_launch_one<BF, T, 1>(grid, st, a);
    case 2: return hs_launch_one<BF, T, 2>(grid, st, a);
    case 3: return hs_launch_one<BF, T, 3>(grid, st, a);
    case 4: return hs_launch_one<BF, T, 4>(grid, st, a);
    default: return fail(AXVS_ERR_ARG, "internal: head-split kernel covers frames of up to 128 keys");
  }
}

#define AXVS_HS_NAME2(bf, tg) launch_hs_bf##bf##_tg##tg
#define AXVS_HS_NAME(bf, tg) AXVS_HS_NAME2(bf, tg)
int AXVS_HS_NAME(AXVS_HS_BF, AXVS_HS_TG)(int T, int nks, unsigned grid, hipStream_t st, const HsArgs& a) {
  constexpr bool BF = AXVS_HS_BF != 0;
  constexpr int T0 = 3 * AXVS_HS_TG + 1;
  if (T == T0) return hs_launch_t<BF, T0>(nks, grid, st, a);
  if (T == T0 + 1) return hs_launch_t<BF, T0 + 1>(nks, grid, st, a);
  if (T == T0 + 2) return hs_launch_t<BF, T0 + 2>(nks, grid, st, a);
  return fail(AXVS_ERR_ARG, "internal: frame count outside this unit");
}

}  // namespace axvs
