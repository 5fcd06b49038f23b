// Host-side plumbing shared by the translation units of libaxvs.so: error string, per-device kernel attributes, the
// packed-weight / workspace views of one trajectory attention, and the launcher of the fused trajectory kernels (whose
// instantiations are compiled one (operand type, frame count) pair per translation unit: axvs_temporal_inst.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/axvs.h"
#include "axvs_common.h"

namespace axvs {

inline thread_local char g_err[512] = "";
// axvs_set_option("train_valu", 1): keep the spatial attention of the training tier on the fp32 VALU kernels (the head_dim 8 / 16
// path) instead of the fp32 MFMA ones -- A/B comparisons and tests of both.  Process-wide, not per thread: autograd runs the
// backward call on its own thread and forward / backward must agree on the kernel family (the statistics travel between them).
inline int g_train_valu = 0;
// The FORWARD GEMMs of the training tier split their operands into three bf16 pieces (fp32 accuracy, 6 MFMAs per product: the
// forward decides the ReLU masks); axvs_set_option("train_exact", 0) makes them two-piece (1.5e-5 per product, 3 MFMAs) like the
// backward ones, 2: three pieces for the input-gradient GEMMs as well -- see axvs_train_gemm.h.  Process-wide like train_valu.
inline int g_train_exact = 1;
// axvs_set_option("train_amp", 1 | 2): the GEMMs of the training tier (forward, input gradients, weight gradients) take ONE 16-bit piece per
// operand (1: bf16, 2: fp16) -- the products torch.autocast gives the reference's nn.Linear layers; 0 (default): split precision
inline int g_train_amp = 0;
// axvs_set_option("train_spatial_wgs", n): workgroups the spatial-attention kernels of the training tier are spread over
inline int g_spatial_wgs = 512;
// axvs_set_option("train_attn_split", 0): the training tier's attention forward on the fp32 MFMA kernel (any axis length) instead of
// the split-precision 16-bit MFMA one (three bf16 pieces per operand, a frame's score tiles in registers; axis length <= 128)
inline int g_train_attn_split = 1;

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: remember (device, kernel) pairs that have
// been raised so a second device in the same process is configured too.  The attribute is always raised to the full 160 KiB,
// never to the size of the launch at hand: the cache below remembers only THAT a kernel was configured, so a shape-dependent
// size would pin the first (possibly smaller) request and fail a later, larger launch.
inline int ensure_max_lds(const void* fn) {
  constexpr int kCap = 256;
  struct Entry { int dev; const void* fn; };
  static thread_local Entry seen[kCap];
  static thread_local int nseen = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return fail(AXVS_ERR_LAUNCH, "hipGetDevice failed");
  for (int i = 0; i < nseen; ++i)
    if (seen[i].fn == fn && seen[i].dev == dev) return AXVS_OK;
  // the dynamic part may use what the kernel's static __shared__ objects leave of the CU's 160 KiB
  hipFuncAttributes fa;
  if (hipFuncGetAttributes(&fa, fn) != hipSuccess) return fail(AXVS_ERR_LAUNCH, "hipFuncGetAttributes failed");
  const int bytes = 160 * 1024 - (int)((fa.sharedSizeBytes + 255) / 256 * 256);
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess)
    return fail(AXVS_ERR_LAUNCH, "hipFuncSetAttribute failed");
  if (nseen < kCap) seen[nseen++] = Entry{dev, fn};
  return AXVS_OK;
}

struct TrajPacked {  // pointers into the packed blob
  u16 *wq, *wk, *wv, *wpq, *wpkv, *wp, *wk2t;     // wk2t: transposed k half of proj_kv (fused temporal kernel, C = 256)
  u16 *wk2n, *wv2h;                               // per-head copies for the reassociated generic temporal half (axvs_misc.h)
  float *bq, *bk, *bv, *bpq, *bpkv, *bp;
  const float *post_ln_g = nullptr, *post_ln_b = nullptr;   // optional LayerNorm on (residual + attention) in the kernel epilogue (not in the blob)
};

struct TrajWs {
  u16 *q16, *k16, *v16, *x16, *o16, *vt16;
  float *q2, *kv2;
};

struct FfnArgs;   // axvs_fused.h
struct OwnQkv;    // axvs_fused.h

// nks = 0: x staged from global (after spatial_attn_kernel); nks > 0: spatial half inside the kernel.  `fa` non-null: the
// layer's FFN rides along (needs nks > 0 and 64-row tiles).
template <bool BF, int T, int MT>
int launch_temporal_n(int nks, const TrajWs& w, const TrajPacked& p, const float* res, float* out, RowMap rm, long long Mp, int N,
                      int L, float scale, hipStream_t st, const FfnArgs* fa, int flags /* bit 0: write-through output rows, bits 1-2: stop after the spatial half / the q,k,v part, bits 4-5: 16-bit output map of the FFN-carrying kernel (kOutF16 / kOutBf16) */,
                      const OwnQkv* oq /* non-null: the kernel computes q, k, v of its own rows first (merged launch) */);

}  // namespace axvs
