// C-ABI entry points of libaxvs.so (see include/axvs.h) and the launch sequences behind them.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

#include "axvs_host.h"
#include "axvs_attn.h"
#include "axvs_cc.h"
#include "axvs_msda.h"
#include "axvs_glue.h"
#include "axvs_lsap.h"
#include "axvs_common.h"
#include "axvs_fused.h"
#include "axvs_ffn_split.h"
#include "axvs_ffn_wide.h"
#include "axvs_gemm.h"
#include "axvs_misc.h"
#include "axvs_gemm_nt.h"

using namespace axvs;

namespace {

// optional per-stage event recording (bench.py / tuning): events[i] is recorded on the stream after stage i
thread_local hipEvent_t* g_prof_events = nullptr;
thread_local int g_prof_cap = 0;
thread_local int g_prof_next = 0;
constexpr int kMaxStages = 32;
thread_local const char* g_stage_names[kMaxStages] = {};
thread_local int g_generic_only = 0;
constexpr int g_attn_waves = 0;
thread_local int* g_status = nullptr;     // axvs_set_status_buffer: word that kernels OR condition bits into (device memory, or pinned host memory)
thread_local volatile int* g_status_host = nullptr;   // the same word when the HOST can read it (pinned host memory): the entry points of the
                                          // axial layer then refuse to run on top of a reported hand-off timeout (status_gate)
thread_local unsigned g_sync_spin_limit = axvs::kSyncSpinLimit;   // option "sync_spin_limit": polls before a hand-off wait gives up (tests shorten it)
thread_local long long g_row_span = 0;   // rows spanned by the layer's row-addressed tensors when their frames are strided (0: natural)
thread_local int g_ffn_wide = 0;         // option "ffn_wide": 0 = 128-row FFN tiles when they save a round of the chip (ffn_wide_pays), 1 = always, 2 = never
constexpr int g_merge_mid = 1;        // option "merge_mid": merged q/k/v + trajectory launch on 32-row tiles (T = 5 .. 8): 1 = while the pass fits one round of the chip, 0 never, 2 always
thread_local int g_no_small_tiles = 0;   // option "no_small_tiles": never use the 16-row trajectory tiles
constexpr int kSmallBelow = 65;          // problems with fewer 64-row tiles than this run the few-rows forms (16-row trajectory tiles, 3-way split q/k/v
                                         // projection, chunk-per-workgroup FFN): their 4x workgroups fit one round of the 256 CUs up to 64 tiles, and from 65 on
                                         // the 64-row forms (merged launch per pass, FFN riding in the width pass) are faster at every T -- round 5 sweep,
                                         // profiles/r5_planner_threshold.txt (128 until then: [1,2,256,48,80] 93.4 -> 79.5 us, [1,5,256,24,40] 94.6 -> 82.8)
constexpr int g_small_below = kSmallBelow;      // option "small_tiles_below" (A/B runs; <= 0 restores the default)
thread_local int g_ffn_split_pairs = 1;  // option "ffn_split_pairs": 65 .. 128 tiles run the chunk-per-workgroup FFN with two chunks per workgroup (0: the one-workgroup-per-tile kernel)
constexpr int g_ffn_split_below = kSmallBelow;  // option "ffn_split_below": the same switch for the stand-alone FFN alone (chunk-per-workgroup form below it)
thread_local int g_spatial_only = 0;     // option "spatial_only": 1 = the fused trajectory kernels return after QK^T / softmax / AV (timing only; outputs unwritten);
                                         // 2 = the merged q/k/v + trajectory kernels return after their q/k/v part (the two-launch kernels treat it as 1)
constexpr int g_no_wt_stores = 0;     // option "no_wt_stores": plain instead of write-through (sc1) stores of inter-kernel tensors (tuning)
thread_local int g_no_ffn_fusion = 0;    // option "no_ffn_fusion": keep the FFN in its own kernel
thread_local int g_no_reassoc = 0;       // option "no_reassoc": generic tier computes k2, v2 = proj_kv(x) for every frame slot (the reference's form)
thread_local int g_ffn_gelu = 0;         // option "ffn_gelu": the layer's FFN activation is exact GELU (F.gelu) instead of ReLU -- set by the
                                         // host module around its calls for activation="gelu" (WC/temporal_attention.py:9-17): the FFN then runs on the
                                         // stand-alone fused kernels' GELU instantiation instead of riding in the width-pass kernel
// option "msda_gemm": the deformable attention's three projections on the 128 x 128 split-precision GEMM of axvs_gemm_nt.h when the
// level set has >= 2048 rows.  4 (the default of rounds 3 - 4): two bf16 pieces for value_proj (its output is rounded to 16 bits anyway) and for the
// offset | weight projection, three pieces (fp32 accuracy) for output_proj, whose result enters the residual stream without a norm;
// 2 (default since the end of round 5) / 3: two / three pieces everywhere; 0: the 64 x 64 kernels of axvs_gemm.h.
// (Two pieces put 5e-6 on a projection; the free-running 16-bit stack's max-norm at BASELINE config 3 is chaotic in its 16-bit roundings either way -- 1.29e-3 with 2,
//  1.38 - 1.48e-3 with 4, relative L2 5.7e-4 for both -- and 2 saves 2.5 % of the module: profiles/r5_planner_threshold.txt.)
constexpr int g_msda_gemm = 2;
constexpr int g_conv_nt128_nchw = 128;  // option "conv_nt128_nchw": the same for NCHW inputs (transposed to token rows first), tiles of the ONE launch over all frames
constexpr int g_conv_nt128_exact = 0;  // option "conv_nt128_exact": 0 = two bf16 pieces per operand (5e-6 of the float64 projection + GroupNorm, 114 against 147 us at [32786 x 256 x 512]), 1 = three (9e-7)
constexpr int g_conv_nt128_splitk = 1024; // option "conv_nt128_splitk": split-K for the NCHW projections with few row tiles and Cin >= this (0: never)
constexpr int g_conv_nt128 = 192;     // option "conv_nt128": token-row 1x1 projections run the 128 x 128 three-piece GEMM from this many tiles per launch on (0: never)
// Merged q/k/v + trajectory launches (temporal_fused_kernel<..., MQ>): one launch per axial pass.  The sibling row tiles of a
// sequence hand K / V^T over inside the launch through arrival counters the CALLER provides (axvs_set_sync_buffer: device words
// that are zero when registered; every launch leaves them zero) -- without a registered buffer the passes run as two launches.
thread_local unsigned* g_sync = nullptr;
thread_local size_t g_sync_words = 0;
thread_local int g_merge_qkv_any = 0;    // option "merge_qkv_any": merged launches at every grid size (A/B; see run_traj)
constexpr int g_qkv_split_upto = 64;  // option "qkv_split_upto": the stand-alone q/k/v kernel runs one workgroup per (tile, q | k | v) up to this many tiles of 64 rows
constexpr int kMergeSmall = 128;         // 16-row-tile passes (T <= 4) of at most this many tiles run merged too (round 5, profiles/r5_merged_16row_tiles.txt)
thread_local int g_merge_small = kMergeSmall;      // option "merge_small": 0 never, 1 at any size, n > 1: passes of at most n tiles of 16 rows, < 0: the default
thread_local int g_out_dtype = 0;        // option "layer_out_dtype": 0 = the layer's output rows are fp32 (the reference's type); 1 / 2 = the kernel that ends the layer
                                         // (norm2 epilogue of the FFN) writes them as f16 / bf16 -- the map a batch-sharded caller gathers over the links
                                         // (BASELINE config 5 is worded "bf16"), written once instead of cast by a second pass
thread_local int g_cc_aspp_affine = 0;   // option "cc_aspp_affine": the ASPP projection's norm is a per-channel affine (norm_fn 'syncbn' in eval mode, folded by the caller into
                                         // aspp_norm_w / aspp_norm_b = scale / shift; 'none' = ones / zeros) instead of the channels-first LayerNorm of the shipped configs
thread_local int g_cc_last_only = 0;     // option "cc_last_heads_only": axvs_cc_module_fwd computes the predictor heads (class logits, mask einsum) of the LAST layer
                                         // only; pred_logits / pred_masks then hold ONE layer.  The reference computes every layer's predictions in eval too
                                         // (CC/...:283-318) and its inference path drops all but the last (maxtron_cc_model.py:301-: aux_outputs are read under
                                         // self.training only): an inference pipeline that does not want them saves 3/4 of the mask einsum's HBM writes
thread_local int g_no_merge_qkv = 0;     // option "no_merge_qkv": keep qkv_fused_kernel + trajectory kernel as two launches (A/B, tests)
thread_local int g_no_attn_fusion = 0;   // option "no_attn_fusion": keep spatial_attn_kernel + temporal kernel separate   // option "attn_waves": cap on waves per attention workgroup (tuning)
   // option "generic_only": 1 = always use the shape-generic v1 kernels

inline void mark(hipStream_t st, const char* name) {
  if (g_prof_next < kMaxStages) g_stage_names[g_prof_next] = name;
  if (g_prof_events && g_prof_next < g_prof_cap) (void)hipEventRecord(g_prof_events[g_prof_next], st);
  ++g_prof_next;
}

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

struct Carver {  // bump allocator over a caller-owned buffer
  char* base;
  size_t off = 0;
  explicit Carver(void* p) : base(static_cast<char*>(p)) {}
  template <class T>
  T* take(size_t n) {
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off = align_up(off + n * sizeof(T));
    return p;
  }
};

// CUs of the current device (the persistent merged launches run one workgroup per CU: their LDS footprint admits no second one)
// The bf16 operand tier sits OUTSIDE the 1e-3 parity bar (2.6e-3 .. 7e-3 against the reference; a bf16 significand has 8 bits) and is not part of the
// default library since round 6: build with -DAXVS_WITH_BF16 to get it (every kernel is a template over the operand type; `kBF` below is the bf16
// arm's template argument, which collapses onto the f16 code -- never reached -- when the tier is not built).
#ifdef AXVS_WITH_BF16
constexpr bool kBF = true;
#else
constexpr bool kBF = false;
#endif
int check_dtype(int dtype) {
  if (dtype == AXVS_F16) return AXVS_OK;
  if (dtype == AXVS_BF16) {
    if (kBF) return AXVS_OK;
    return fail(AXVS_ERR_ARG, "the bf16 operand tier is not built into this library (it does not hold the 1e-3 parity bar; fp16 operands run at the same rate and do): rebuild with AXVS_WITH_BF16=1");
  }
  return fail(AXVS_ERR_ARG, "unknown dtype %d", dtype);
}

int cu_count() {
  static thread_local int dev_seen = -1, cus = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (dev != dev_seen) {
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, dev) != hipSuccess) return 0;
    cus = pr.multiProcessorCount;
    dev_seen = dev;
  }
  return cus;
}

int check_cfg(int C, int heads) {
  if (C <= 0 || heads <= 0 || C % heads != 0) return fail(AXVS_ERR_ARG, "C=%d must be a positive multiple of heads=%d", C, heads);
  if (C % 32 != 0) return fail(AXVS_ERR_ARG, "C=%d must be a multiple of 32", C);
  if (C / heads > 32) return fail(AXVS_ERR_ARG, "head_dim=%d > 32 is not supported yet", C / heads);
  return AXVS_OK;
}

// Fail loudly (round 5): a hand-off wait of a merged launch that ran out has set AXVS_STATUS_SYNC_TIMEOUT; that launch's outputs are
// garbage and its arrival counters are left non-zero.  When the status word is host-readable the NEXT call of an axial-layer entry
// point on this thread returns AXVS_ERR_STATE instead of computing on top of it (no synchronisation: the word is read as it is).
int status_gate() {
  if (g_status_host != nullptr && (*g_status_host & AXVS_STATUS_SYNC_TIMEOUT))
    return fail(AXVS_ERR_STATE, "an earlier merged q/k/v + trajectory launch gave up waiting for its sibling row tiles (status bit 2, "
                                "AXVS_STATUS_SYNC_TIMEOUT): its outputs are invalid and its sync words are not zero.  Synchronise, zero the sync words "
                                "(axvs_set_sync_buffer contract), clear the status word, then call again -- or run two launches per pass "
                                "(axvs_set_sync_buffer(NULL, 0) / option no_merge_qkv)");
  return AXVS_OK;
}

// ---------------- packed weights ----------------
TrajPacked carve_traj(Carver& c, int C, int heads) {
  const size_t Cp = (size_t)heads * 32;
  TrajPacked t{};
  t.wq = c.take<u16>(Cp * C);
  t.wk = c.take<u16>(Cp * C);
  t.wv = c.take<u16>(Cp * C);
  t.wpq = c.take<u16>(Cp * Cp);
  t.wpkv = c.take<u16>(2 * Cp * Cp);
  t.wk2t = c.take<u16>(Cp * Cp);
  t.wk2n = c.take<u16>(Cp * Cp);
  t.wv2h = c.take<u16>(Cp * Cp);
  t.wp = c.take<u16>((size_t)C * Cp);
  t.bq = c.take<float>(Cp);
  t.bk = c.take<float>(Cp);
  t.bv = c.take<float>(Cp);
  t.bpq = c.take<float>(Cp);
  t.bpkv = c.take<float>(2 * Cp);
  t.bp = c.take<float>(C);
  return t;
}

struct LayerPacked {
  TrajPacked th, tw;
  u16 *w1, *w2;
  float *b1, *b2, *g1, *be1, *g2, *be2;
};

LayerPacked carve_ffn(Carver& c, int C, int F) {   // norm1 / linear1 / linear2 / norm2 only (th, tw unused)
  LayerPacked l{};
  l.w1 = c.take<u16>((size_t)F * C);
  l.w2 = c.take<u16>((size_t)F * C);
  l.b1 = c.take<float>(F);
  l.b2 = c.take<float>(C);
  l.g1 = c.take<float>(C);
  l.be1 = c.take<float>(C);
  l.g2 = c.take<float>(C);
  l.be2 = c.take<float>(C);
  return l;
}

LayerPacked carve_layer(Carver& c, int C, int heads, int F) {
  LayerPacked l;
  l.th = carve_traj(c, C, heads);
  l.tw = carve_traj(c, C, heads);
  l.w1 = c.take<u16>((size_t)F * C);
  l.w2 = c.take<u16>((size_t)F * C);
  l.b1 = c.take<float>(F);
  l.b2 = c.take<float>(C);
  l.g1 = c.take<float>(C);
  l.be1 = c.take<float>(C);
  l.g2 = c.take<float>(C);
  l.be2 = c.take<float>(C);
  return l;
}

template <bool BF>
void pack_w(const float* W, u16* out, PackDim nd, PackDim kd, hipStream_t st, int n_off = 0, int n_total = 0) {
  long long total = (long long)nd.padded * kd.padded;
  hipLaunchKernelGGL((pack_weight_kernel<BF>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, W, out, nd, kd, n_off,
                     n_total ? n_total : nd.padded);
}
template <bool BF>
void pack_w3(const float* W, u16* out, PackDim nd, PackDim kd, hipStream_t st, int n_off = 0, int n_total = 0) {   // split precision
  long long total = (long long)nd.padded * kd.padded;
  hipLaunchKernelGGL((pack_weight_split3_kernel<BF>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, W, out, nd, kd, n_off,
                     n_total ? n_total : nd.padded);
}
void pack_b(const float* b, float* out, PackDim nd, hipStream_t st) {
  hipLaunchKernelGGL(pack_bias_kernel, dim3((nd.padded + 255) / 256), dim3(256), 0, st, b, out, nd);
}

template <bool BF>
void pack_ffn(const float* n1w, const float* n1b, const float* l1w, const float* l1b, const float* l2w, const float* l2b,
              const float* n2w, const float* n2b, const LayerPacked& l, int C, int F, hipStream_t st) {
  PackDim plainC{C, C, 0, 0, 0}, plainF{F, F, 0, 0, 0};
  pack_w<BF>(l1w, l.w1, plainF, plainC, st);
  pack_w<BF>(l2w, l.w2, plainC, plainF, st);
  pack_b(l1b, l.b1, plainF, st);
  pack_b(l2b, l.b2, plainC, st);
  pack_b(n1w, l.g1, plainC, st);
  pack_b(n1b, l.be1, plainC, st);
  pack_b(n2w, l.g2, plainC, st);
  pack_b(n2b, l.be2, plainC, st);
}

template <bool BF>
void pack_traj(const AxvsTrajParams& p, const TrajPacked& t, int C, int heads, hipStream_t st) {
  const int d = C / heads, Cp = heads * 32;
  // The kernels store q, k and x (the spatial-attention output) with the 32 channels of a head block in perm32 order
  // (16-byte stores per lane).  q.k is invariant to a common permutation of d; proj_q / proj_kv consume x, so their K
  // columns are packed in the same order.
  PackDim plainC{C, C, 0, 0, 0}, headC{C, Cp, heads, d, 0}, head2C{2 * C, 2 * Cp, heads, d, 0}, headCp{C, Cp, heads, d, 1};
  pack_w<BF>(p.q_w, t.wq, headC, plainC, st);
  pack_w<BF>(p.k_w, t.wk, headC, plainC, st);
  pack_w<BF>(p.v_w, t.wv, headC, plainC, st);
  pack_w<BF>(p.proj_q_w, t.wpq, headC, headCp, st);
  pack_w<BF>(p.proj_kv_w, t.wpkv, head2C, headCp, st);
  if (d == 32) {
    const dim3 pg((unsigned)(((long long)heads * C * 32 + 255) / 256));
    hipLaunchKernelGGL((pack_wk2t_kernel<BF>), pg, dim3(256), 0, st, p.proj_kv_w, t.wk2t, C, heads);
    hipLaunchKernelGGL((pack_wk2n_kernel<BF>), pg, dim3(256), 0, st, p.proj_kv_w, t.wk2n, C, heads);
    hipLaunchKernelGGL((pack_wv2h_kernel<BF>), pg, dim3(256), 0, st, p.proj_kv_w, t.wv2h, C, heads);
  }
  pack_w<BF>(p.proj_w, t.wp, plainC, headC, st);
  pack_b(p.q_b, t.bq, headC, st);
  pack_b(p.k_b, t.bk, headC, st);
  pack_b(p.v_b, t.bv, headC, st);
  pack_b(p.proj_q_b, t.bpq, headC, st);
  pack_b(p.proj_kv_b, t.bpkv, head2C, st);
  pack_b(p.proj_b, t.bp, plainC, st);
}

// ---------------- one trajectory attention over sequence-ordered rows ----------------
// lean: the fully fused tier only round-trips q, k and V^T (x, the T-expanded attention output, stays in LDS)
// rows of the q/k/v row space of `M` token rows in frames of L rows: the fused trajectory tier pads every frame to a multiple of 16
// rows (RowMap, "padded frames"), so that 16-row MFMA tiles never straddle frames
inline int pad16(int L) { return (L + 15) & ~15; }
inline long long padded_rows(long long M, int L) { return L > 0 && L % 16 ? M / L * pad16(L) : M; }

// Mq: capacity of q16 / k16 / vt16 in rows (>= Mp: padded_rows of the longest frame the caller will run)
TrajWs carve_traj_ws(Carver& c, long long Mp, int T, int heads, bool lean = false, long long Mq = 0) {
  const size_t Cp = (size_t)heads * 32;
  if (Mq < Mp) Mq = Mp;
  TrajWs w{};
  w.q16 = c.take<u16>(Cp * Mq);
  w.k16 = c.take<u16>(Cp * Mq);
  w.vt16 = c.take<u16>(2 * Cp * Mq);      // block-transposed V (frames padded to a multiple of 32 keys: at most 2x)
  if (lean) return w;
  w.v16 = c.take<u16>(Cp * Mp);
  w.x16 = c.take<u16>(Cp * Mp * T);
  w.o16 = c.take<u16>(Cp * Mp);
  w.q2 = c.take<float>(Cp * Mp);
  // (k2 | v2) of every frame slot, or -- reassociated temporal half, T > 5 -- u and z: [Mp][heads * Cp] fp32 each
  w.kv2 = c.take<float>(((size_t)2 * Cp * T > (size_t)2 * heads * Cp ? (size_t)2 * Cp * T : (size_t)2 * heads * Cp) * Mp);
  return w;
}

// Full fusion (spatial half inside the temporal kernel, x never leaves LDS) needs: the fused kernels, no attention-map
// output, 8..128 keys per frame (frames are padded to multiples of 16 rows in the q/k/v row space, V^T to 32-key steps: within 2x of
// the padded rows for every L; below 8 keys the padding would more than double the work).  Any axis length: row tiles are cut
// per sequence, partial key tiles are masked.
// Frame counts the fused trajectory kernels exist for: T <= 8 (64- / 32-row tiles), and 9 .. 12 on 16-row tiles (x tile T * 8 KiB)
// for problems with few rows -- whole-video cross-clip inference with up to 12 clips (Q * Tc rows per video).
// (The choice depends on T alone, not on the number of rows of the call: the fused and the generic tier differ at the 16-bit level,
//  and a clip's result must not depend on how many other clips share its batch -- batch sharding is bit-exact.)
bool fused_frames(int T, long long /*rows*/) { return T <= 8 || (T <= 12 && !g_no_small_tiles); }
bool can_fuse_attn(int C, int heads, int T, int L, bool want_attn, long long rows) {
  return !g_generic_only && !g_no_attn_fusion && C == 256 && heads == 8 && fused_frames(T, rows) && !want_attn && L >= 8 && L <= 128;
}
// The FFN rides in the width-pass kernel only when that kernel has more than 64 tiles (kSmallBelow): with fewer 64-row tiles every
// workgroup's private 1 MB FFN weight stream is pure latency (43 us per pass whether 16 or 64 workgroups run), and a 16-row
// trajectory kernel (4x the workgroups) + the stand-alone FFN kernel is faster (BASELINE config 3: res4 / res5 levels).
inline int small_below(int /*T*/) { return g_small_below; }
bool can_fuse_ffn_into_pass(int T, int F, long long M) {
  return !g_no_ffn_fusion && !g_ffn_gelu && T <= 4 && F % 256 == 0 && F <= 4096 && (M >= (long long)small_below(T) * 64 || g_no_small_tiles);
}
// (activation = gelu: the stand-alone fused FFN kernels have a GELU instantiation; only the width-pass kernel does not carry it)
bool ffn_kernel_is_fused(int C, int heads, int F) { return !g_generic_only && C == 256 && heads == 8 && F % 256 == 0 && F <= 4096; }
// few rows: one workgroup per (64-row tile, 256-unit chunk of the hidden layer) + a row-wise finishing kernel (axvs_ffn_split.h);
// bit-identical to the one-workgroup-per-tile kernels, so the row count may decide
// 65 .. 88 tiles (round 5): the same kernel with TWO consecutive chunks per workgroup -- tiles x F/512 workgroups, still one round of the chip -- where the
// one-workgroup-per-tile kernel leaves half the CUs idle behind a private 1 MB stream (only reached when the FFN does not ride in the width pass: T >= 5, GELU)
int ffn_split_mode(int C, int heads, int F, long long M) {      // 0: no split, 1: one chunk per workgroup, 2: two
  if (!ffn_kernel_is_fused(C, heads, F) || g_no_small_tiles || F < 512) return 0;
  if (M < (long long)g_ffn_split_below * 64) return 1;
  return (g_ffn_split_pairs && F % 512 == 0 && M <= 88 * 64) ? 2 : 0;      // measured: -2.4 us at 75 tiles, -0.5 .. -0.9 at 80 .. 84, +0.9 at 96, +4.5 at 128 (partials + finishing kernel)
}
bool ffn_split_applies(int C, int heads, int F, long long M) { return ffn_split_mode(C, heads, F, M) != 0; }

// 64-row tiles (MT = 4) of the fused trajectory kernel: T <= 4, and either the FFN rides along or there are enough tiles to
// fill the chip (few tiles take 16-row tiles: 4x the workgroups) -- the choice launch_temporal makes
bool traj_mt4(int T, long long tiles64, bool with_ffn) { return T <= 4 && (with_ffn || tiles64 >= small_below(T) || g_no_small_tiles); }
long long traj_tiles64(long long Mp, int N) { return (Mp / N) * ((N + 63) / 64); }
// what one axial layer's launch sequence touches in the workspace (the same predicates run_traj / run_ffn dispatch on)
struct LayerPlan {
  bool lean_traj;     // both passes fully fused
  bool need_buf2;     // the width pass writes rows for a separate FFN launch
  bool need_ffn_tmp;  // generic FFN (LayerNorm / GEMM / GEMM / LayerNorm): fp32 scratch + 16-bit y and h
  bool need_ffn_part; // chunk-per-workgroup FFN for few rows: [F/256][M][256] fp32 partial outputs
};
LayerPlan plan_layer(int B, int T, int H, int W, int C, int heads, int F, bool want_attn) {
  LayerPlan p;
  const long long rows = (long long)B * T * H * W;
  p.lean_traj = can_fuse_attn(C, heads, T, H, want_attn, rows) && can_fuse_attn(C, heads, T, W, want_attn, rows);
  p.need_buf2 = !(can_fuse_attn(C, heads, T, W, want_attn, rows) && can_fuse_ffn_into_pass(T, F, rows));
  p.need_ffn_tmp = p.need_buf2 && !ffn_kernel_is_fused(C, heads, F);
  p.need_ffn_part = p.need_buf2 && ffn_split_applies(C, heads, F, rows);
  return p;
}

RowMap identity_map(long long rows) {
  int n = (int)(rows > 0 ? rows : 1);
  return RowMap{n, n, 1, 0, 0, 1, 0};
}

template <bool BF, int NKS>
int launch_attn(const TrajWs& w, float* attn, int S, int N, int T, int L, int heads, long long Mp, hipStream_t st) {
  // K and V of a (sequence, head) are staged in LDS, as many frames at a time as fit (whole-video cross-clip inference: T = clips)
  const size_t per_frame = (size_t)NKS * 32 * 32 * 2 * sizeof(u16);
  const int tch = (int)((150 * 1024) / per_frame) < T ? (int)((150 * 1024) / per_frame) : T;
  const size_t lds = per_frame * tch;
  if (int rc = ensure_max_lds(reinterpret_cast<const void*>(&spatial_attn_kernel<BF, NKS>))) return rc;
  const int wcap = g_attn_waves > 0 ? g_attn_waves : 8;
  const int nwaves = (N + 31) / 32 >= wcap ? wcap : (N + 31) / 32;  // 32 queries per wave, at most `wcap` waves
  dim3 grid((N + 32 * nwaves - 1) / (32 * nwaves), heads, S);
  hipLaunchKernelGGL((spatial_attn_kernel<BF, NKS>), grid, dim3(64 * nwaves), lds, st, w.q16, w.k16, w.v16, w.x16, attn, N, T, L,
                     heads, Mp, tch);
  return AXVS_OK;
}

// nks = 0: x staged from global (after spatial_attn_kernel); nks > 0: spatial half inside the kernel
template <bool BF>
int launch_temporal(const TrajWs& w, const TrajPacked& p, const float* res, float* out, RowMap rm, long long Mp, int N, int L, int T,
                    float scale, hipStream_t st, int nks = 0, const FfnArgs* fa = nullptr, const OwnQkv* oq = nullptr) {
  // output rows are addressed through the RowMap: the largest byte offset is that of the natural [rows, 256] fp32 tensor
  const int wt = ((!g_no_wt_stores && (g_row_span ? g_row_span : Mp) * 256 * 4 < (1ll << 32)) ? 1 : 0) | (g_spatial_only && nks > 0 ? (oq ? g_spatial_only : 1) << 1 : 0) |
                 (fa != nullptr && g_out_dtype ? (g_out_dtype == 1 ? kOutF16 : kOutBf16) : 0);
  // few rows (cross-clip queries: 512 per video): 16-row tiles give 4x the workgroups -- the spatial half is per-query work
  const long long tiles64 = nks > 0 ? traj_tiles64(Mp, N) : (Mp + 63) / 64;
  if (oq && !(nks > 0 && T <= 8)) return fail(AXVS_ERR_ARG, "internal: own q,k,v need the in-kernel spatial half and T <= 8");
  if (fa == nullptr && (tiles64 < small_below(T) || T > 8) && !g_no_small_tiles) {       // (T > 8 exists on 16-row tiles only: fused_frames)
    switch (T) {
      case 1: return launch_temporal_n<BF, 1, 1>(nks, w, p, res, out, rm, Mp, N, L, scale, st, fa, wt, oq);
      case 2: return launch_temporal_n<BF, 2, 1>(nks, w, p, res, out, rm, Mp, N, L, scale, st, fa, wt, oq);
      case 3: return launch_temporal_n<BF, 3, 1>(nks, w, p, res, out, rm, Mp, N, L, scale, st, fa, wt, oq);
      case 4: return launch_temporal_n<BF, 4, 1>(nks, w, p, res, out, rm, Mp, N, L, scale, st, fa, wt, oq);
      case 5: return launch_temporal_n<BF, 5, 1>(nks, w, p, res, out, rm, Mp, N, L, scale, st, fa, wt, oq);
      case 6: return launch_temporal_n<BF, 6, 1>(nks, w, p, res, out, rm, Mp, N, L, scale, st, fa, wt, oq);
      case 7: return launch_temporal_n<BF, 7, 1>(nks, w, p, res, out, rm, Mp, N, L, scale, st, fa, wt, oq);
      case 8: return launch_temporal_n<BF, 8, 1>(nks, w, p, res, out, rm, Mp, N, L, scale, st, fa, wt, oq);
      case 9: return launch_temporal_n<BF, 9, 1>(nks, w, p, res, out, rm, Mp, N, L, scale, st, fa, wt, oq);
      case 10: return launch_temporal_n<BF, 10, 1>(nks, w, p, res, out, rm, Mp, N, L, scale, st, fa, wt, oq);
      case 11: return launch_temporal_n<BF, 11, 1>(nks, w, p, res, out, rm, Mp, N, L, scale, st, fa, wt, oq);
      case 12: return launch_temporal_n<BF, 12, 1>(nks, w, p, res, out, rm, Mp, N, L, scale, st, fa, wt, oq);
      default: break;
    }
  }
  switch (T) {
    case 1: return launch_temporal_n<BF, 1, 4>(nks, w, p, res, out, rm, Mp, N, L, scale, st, fa, wt, oq);
    case 2: return launch_temporal_n<BF, 2, 4>(nks, w, p, res, out, rm, Mp, N, L, scale, st, fa, wt, oq);
    case 3: return launch_temporal_n<BF, 3, 4>(nks, w, p, res, out, rm, Mp, N, L, scale, st, fa, wt, oq);
    case 4: return launch_temporal_n<BF, 4, 4>(nks, w, p, res, out, rm, Mp, N, L, scale, st, fa, wt, oq);
    case 5: return launch_temporal_n<BF, 5, 2>(nks, w, p, res, out, rm, Mp, N, L, scale, st, fa, wt, oq);
    case 6: return launch_temporal_n<BF, 6, 2>(nks, w, p, res, out, rm, Mp, N, L, scale, st, fa, wt, oq);     // x tile: T * 16 KiB of LDS
    case 7: return launch_temporal_n<BF, 7, 2>(nks, w, p, res, out, rm, Mp, N, L, scale, st, fa, wt, oq);
    case 8: return launch_temporal_n<BF, 8, 2>(nks, w, p, res, out, rm, Mp, N, L, scale, st, fa, wt, oq);
    default: return fail(AXVS_ERR_ARG, "fused temporal kernel supports T <= 8");
  }
}

// q/k/v inputs are fp32 token rows addressed through `rm`; `qk_add` (nullable) is added to the q and k inputs.
// Result (+ bias, + optional residual `res`) goes to fp32 rows of `out` through `rm`.
template <bool BF>
int run_traj(const float* qsrc, const float* ksrc, const float* vsrc, const float* qk_add, const float* res, float* out,
             float* attn, const TrajPacked& p, const TrajWs& w, RowMap rm, int S, int T, int L, int C, int heads,
             hipStream_t st, int pass = 0, const FfnArgs* ffn = nullptr, float* ffn_out = nullptr, bool* ffn_done = nullptr,
             const PosGen* posgen = nullptr,
             bool may_merge = false /* the caller's sequences may use the registered sync words (one trajectory call at a time per buffer) */) {
  static const char* const kNames[3][8] = {
      {"qkv_proj", "spatial_attn", "proj_q", "proj_kv", "temporal_attn", "proj", "temporal_fused", "traj_fused"},
      {"h.qkv_proj", "h.spatial_attn", "h.proj_q", "h.proj_kv", "h.temporal_attn", "h.proj", "h.temporal_fused", "h.traj_fused"},
      {"w.qkv_proj", "w.spatial_attn", "w.proj_q", "w.proj_kv", "w.temporal_attn", "w.proj", "w.temporal_fused", "w.traj_fused"}};
  const char* const* nm = kNames[pass];
  if (may_merge)      // never compute on top of a reported hand-off timeout (host-readable status word; no synchronisation)
    if (int rc = status_gate()) return rc;
  const int Lreal = L;
  const int Cp = heads * 32, d = C / heads;
  const long long Mreal = (long long)S * T * L;
  if (Mreal * T > 2147483647LL / 4) return fail(AXVS_ERR_ARG, "too many tokens for 32-bit row indices");
  const float scale = 1.0f / sqrtf((float)d);
  const float kLog2e = 1.4426950408889634f;

  const bool fuse_attn = qsrc == ksrc && vsrc == qsrc && can_fuse_attn(C, heads, T, L, attn != nullptr, Mreal);
  const bool with_ffn = fuse_attn && ffn != nullptr && can_fuse_ffn_into_pass(T, ffn->F, Mreal);
  // The fused tier runs in the PADDED row space (RowMap): frames of roundup16(L) rows, the last ones of each frame clamped copies
  // that are computed and never stored -- every 16-row MFMA tile then lies inside one frame, K / V^T are stored 16 / 8 bytes per lane
  // and the merged launch applies for any frame length (the shipped VIPSeg maps: 49 x 85, 25 x 43).  Every other tier is dense.
  if (fuse_attn && L % 16 != 0) {
    L = pad16(L);
    rm.Lv = Lreal;
    rm.L = L;
    rm.N = T * L;
  }
  const int N = T * L;
  const long long Mp = (long long)S * N;
  const int M = (int)Mp;
  const int nks_fused = (L + 31) / 32;
  // one launch per pass: the trajectory kernel computes q, k, v of its own rows (OwnQkv).  Needs the 64-row fused kernels with
  // T <= 4, frames of a multiple of 16 keys (16-byte / 8-byte V^T stores) and at most 96 (register budget), byte offsets of K / V^T
  // below 4 GiB (buffer addressing), one registered arrival counter per sequence.  Bit-identical to the two-launch form.
  // Frames of 64 keys (a row tile IS a frame: its own K / V^T fragments never leave the registers, MQ = 2) gain at every size
  // measured (4 - 9 % from [1,2,256,64,64] to [8,4,256,64,64]); other frame lengths (MQ = 1) gain while the grid stays within
  // ~2 rounds of the chip (-3.5 % at 576 tiles, -7.5 % at 240) and LOSE beyond (+2.5 % at 1152 tiles, +8 % at 4608: sibling
  // tiles start staggered there and every tile waits for the last one) -- option "merge_qkv_any" lifts the limit for A/B runs.
  // Problems with few rows run 16-row tiles (launch_temporal: fewer than kSmallBelow tiles of 64 rows and no FFN riding along -- pyramid
  // levels of 32 x 32 and below, the cross-clip queries).  Their merged form exists (MQ = 1 on 16-row tiles, bit-identical) but
  // every 16-row workgroup then streams the 384 KB of q/k/v weights itself, which costs what the launch and the q round trip save:
  // layer at [1,4,256,32,32] 59.0 vs 59.2 us, [1,4,256,16,16] 53.1 vs 56.0, [3,4,256,16,32] 105.3 vs 97.8 (768 tiles: siblings
  // start staggered), cross-clip module 236.1 vs 237.3, BASELINE config 3 0.997 vs 0.989 ms -- off unless option "merge_small" (rounds 3 - 4).
  // Round 5: with the few-rows forms ending at 64 tiles of 64 rows (kSmallBelow) every 16-row grid fits one round of the chip and the case that lost is gone:
  // back to back on warm weights the merged form gains at every size (-8 % per layer at 32 tiles of 16 rows, -7 % at 64, -3 % at 128, -1.5 % at 256), but in a
  // stack of layers with their own, cold weights only up to 128 tiles (-3 %; +3 % at 256: 256 workgroups x 384 KB of q/k/v weights from HBM) -- merged up to
  // kMergeSmall tiles per pass: BASELINE config 3 0.878 -> 0.864 ms, cross-clip module 226 -> 224.6 us (profiles/r5_merged_16row_tiles.txt).
  const bool mt4 = traj_mt4(T, traj_tiles64(Mp, N), with_ffn);
  const bool own_frame = mt4 && L == 64 && T >= 2;
  const long long tiles = traj_tiles64(Mp, N);
  const int tps = (N + 63) / 64;
  // 32-row tiles (5 .. 8 frames per clip, more than 64 tiles of 64 rows; Tube-Link's T = 5 levels): the merged form exists too (MQ = 1, round 5).  Their x tile
  // (T * 16 KiB) leaves room for ONE workgroup per CU, so the siblings of a hand-off start together only while the pass fits one round of the chip: -4 .. -9 % per
  // layer up to 256 tiles, +4 .. +15 % beyond (profiles/r5_merged_32row_tiles.txt) -- merged iff tiles <= CUs (option "merge_mid": 0 never, 2 always).
  const bool mt2 = T > 4 && T <= 8 && (tiles >= small_below(T) || g_no_small_tiles);
  const bool mid_ok = mt2 && g_merge_mid && (g_merge_mid == 2 || (long long)S * ((N + 31) / 32) <= cu_count());
  const bool merge = may_merge && fuse_attn && !g_generic_only && !g_no_merge_qkv &&
                     g_sync != nullptr && (size_t)S <= g_sync_words && (T <= 4 || mt2) && L % 16 == 0 && nks_fused <= (mt4 || mt2 ? 3 : 4) &&
                     2 * (long long)Cp * Mp * 2 < (1ll << 32) &&
                     (!mt4 || own_frame || tiles <= 640 || g_merge_qkv_any) && (mt4 || mid_ok || (!mt2 && g_merge_small && (g_merge_small == 1 || (long long)S * ((N + 15) / 16) <= g_merge_small)));
  if (merge) {
    const OwnQkv oq{qsrc, qk_add, posgen ? *posgen : PosGen{}, p.wq, p.wk, p.wv, p.bq, p.bk, p.bv, scale * kLog2e, g_sync, g_status, g_sync_spin_limit};
    int rc = launch_temporal<BF>(w, p, res, with_ffn ? ffn_out : out, rm, Mp, N, L, T, scale, st, nks_fused, with_ffn ? ffn : nullptr, &oq);
    if (rc != AXVS_OK) return rc;
    if (with_ffn) *ffn_done = true;
    mark(st, with_ffn ? "w.qkv+traj+ffn" : pass == 1 ? "h.qkv+traj" : pass == 2 ? "w.qkv+traj" : "qkv+traj");
    return AXVS_OK;
  }
  // q, k, v projections -> blocked 16-bit, q pre-multiplied by scale*log2(e) for the exp2 softmax
  if (!g_generic_only && C == 256 && heads == 8 && qsrc == ksrc) {
    if (int rc = ensure_max_lds(reinterpret_cast<const void*>(&qkv_fused_kernel<BF>))) return rc;
    // the fused kernel reads the value rows from the same tensor as the q/k rows (+ optional additive term)
    if (vsrc == qsrc) {
      // (the V^T padding keys of frames that are not multiples of 32 keys are cleared by the kernel itself)
      const unsigned qtiles = (unsigned)((Mp + 63) / 64);
      // few tiles (cross-clip queries): one workgroup per (tile, q | k | v) -- a third of the weight stream each
      hipLaunchKernelGGL((qkv_fused_kernel<BF>), dim3(qtiles, ((int)qtiles <= g_qkv_split_upto && !g_no_small_tiles) ? 3 : 1), dim3(512), kQkvLdsBytes, st, qsrc, qk_add, rm,
                         p.wq, p.wk, p.wv, p.bq, p.bk, p.bv, w.q16, w.k16, w.v16, Mp, scale * kLog2e,
                         fuse_attn ? w.vt16 : (u16*)nullptr, N, L, T, nks_fused, posgen ? *posgen : PosGen{},
                         (!g_no_wt_stores && 2 * (long long)Cp * Mp * 2 < (1ll << 32)) ? 1 : 0, g_status);
      goto qkv_done;
    }
  }
  {
    if (posgen) return fail(AXVS_ERR_ARG, "internal: generated positions need the fused QKV kernel");
    ALoadRowsF32<BF> aq{qsrc, qk_add, rm, M, C}, ak{ksrc, qk_add, rm, M, C}, av{vsrc, nullptr, rm, M, C};
    launch_gemm<BF>(aq, p.wq, EpiBlocked16<BF>{w.q16, Mp, p.bq, scale * kLog2e, Cp, 0}, M, Cp, C, st);
    launch_gemm<BF>(ak, p.wk, EpiBlocked16<BF>{w.k16, Mp, p.bk, 1.f, 0, 0}, M, Cp, C, st);
    launch_gemm<BF>(av, p.wv, EpiBlocked16<BF>{w.v16, Mp, p.bv, 1.f, 0, 0}, M, Cp, C, st);
  }
qkv_done:
  mark(st, nm[0]);
  if (fuse_attn) {
    // the layer's FFN can ride along (64-row tiles, LDS budget): `out` then receives norm2(FFN(norm1(...)))
    int rc = launch_temporal<BF>(w, p, res, with_ffn ? ffn_out : out, rm, Mp, N, L, T, scale, st, nks_fused, with_ffn ? ffn : nullptr);
    if (rc != AXVS_OK) return rc;
    if (with_ffn) *ffn_done = true;
    mark(st, with_ffn ? "w.traj_fused+ffn" : nm[7]);
    return AXVS_OK;
  }

  // spatial half
  int nks = (L + 31) / 32, rc;
  switch (nks) {
    case 1: rc = launch_attn<BF, 1>(w, attn, S, N, T, L, heads, Mp, st); break;
    case 2: rc = launch_attn<BF, 2>(w, attn, S, N, T, L, heads, Mp, st); break;
    case 3: rc = launch_attn<BF, 3>(w, attn, S, N, T, L, heads, Mp, st); break;
    case 4: rc = launch_attn<BF, 4>(w, attn, S, N, T, L, heads, Mp, st); break;
    case 5: rc = launch_attn<BF, 5>(w, attn, S, N, T, L, heads, Mp, st); break;
    case 6: rc = launch_attn<BF, 6>(w, attn, S, N, T, L, heads, Mp, st); break;
    case 7: rc = launch_attn<BF, 7>(w, attn, S, N, T, L, heads, Mp, st); break;
    case 8: rc = launch_attn<BF, 8>(w, attn, S, N, T, L, heads, Mp, st); break;
    default: {   // more than 256 keys per frame (full T*H*W trajectory attention): chunked keys, online softmax
      if (attn != nullptr) return fail(AXVS_ERR_ARG, "attention maps are not available for frames of more than 256 keys (L=%d)", L);
      if (int rc2 = ensure_max_lds(reinterpret_cast<const void*>(&spatial_attn_long_kernel<BF>))) return rc2;
      const int nwaves = (N + 31) / 32 >= 8 ? 8 : (N + 31) / 32;
      dim3 grid((N + 32 * nwaves - 1) / (32 * nwaves), heads, S);
      hipLaunchKernelGGL((spatial_attn_long_kernel<BF>), grid, dim3(64 * nwaves), (size_t)2 * 256 * 32 * sizeof(u16), st, w.q16, w.k16, w.v16,
                         w.x16, N, T, L, heads, Mp);
      rc = AXVS_OK;
    }
  }
  if (rc != AXVS_OK) return rc;
  mark(st, nm[1]);

  // temporal half + output projection + residual
  if (!g_generic_only && C == 256 && heads == 8 && fused_frames(T, Mp)) {
    rc = launch_temporal<BF>(w, p, res, out, rm, Mp, N, L, T, scale, st);
    if (rc != AXVS_OK) return rc;
    mark(st, nm[6]);
    return AXVS_OK;
  }
  ALoadBlocked<BF> adiag{w.x16, Mp * T, M, T, N, L};
  launch_gemm<BF>(adiag, p.wpq, EpiRowsF32{w.q2, nullptr, p.bpq, identity_map(Mp), Cp, scale}, M, Cp, Cp, st);
  mark(st, nm[2]);
  if (!g_generic_only && !g_no_reassoc && C == 256 && heads == 8 && T >= 12) {   // (below ~12 frames the per-head GEMMs cost more than they save)
    // Reassociated (see temporal_fused_kernel): proj_kv is applied to u_h = Wk2_h^T q2_h and z_h = sum_f a_f x_f instead of to
    // every frame slot of x -- 2 C^2 instead of 2 T C^2 MACs per token, and no [T*M, 2C] tensor.  Whole-video cross-clip
    // inference runs T = number of clips (tens): this is what keeps the temporal half linear in T.
    float* U = w.kv2;
    float* Z = w.kv2 + (size_t)Mp * heads * Cp;
    {
      GemmBatch<ALoadRowsLd<BF>, EpiRowsF32, 8> gb;
      for (int h = 0; h < 8; ++h) {
        gb.al[h] = ALoadRowsLd<BF>{w.q2, Cp, h * 32, M};
        gb.W[h] = p.wk2n + (size_t)h * Cp * 32;
        gb.epi[h] = EpiRowsF32{U + h * Cp, nullptr, nullptr, identity_map(Mp), heads * Cp, 1.f};
      }
      launch_gemm_batched<BF>(gb, M, Cp, 32, st);
    }
    hipLaunchKernelGGL((temporal_stream_kernel<BF>), dim3((unsigned)((Mp + 3) / 4)), dim3(256), 0, st, (const float*)U, (const u16*)w.x16, Z,
                       (long long)M, Mp, T);
    mark(st, nm[3]);
    {
      GemmBatch<ALoadRowsLd<BF>, EpiBlocked16<BF>, 8> gb;
      for (int h = 0; h < 8; ++h) {
        gb.al[h] = ALoadRowsLd<BF>{Z, heads * Cp, h * Cp, M};
        gb.W[h] = p.wv2h + (size_t)h * Cp * 32;
        gb.epi[h] = EpiBlocked16<BF>{w.o16, Mp, p.bpkv + Cp + h * 32, 1.f, 0, 0};
        gb.epi[h].n_off = h * 32;
      }
      launch_gemm_batched<BF>(gb, M, 32, Cp, st);
    }
    mark(st, nm[4]);
  } else {
  ALoadBlocked<BF> aall{w.x16, Mp * T, M * T, 0, 1, 1};
  launch_gemm<BF>(aall, p.wpkv, EpiRowsF32{w.kv2, nullptr, p.bpkv, identity_map(Mp * T), 2 * Cp, 1.f}, M * T, 2 * Cp, Cp, st);
  mark(st, nm[3]);
  {
    long long threads = Mp * heads * 8;
    hipLaunchKernelGGL((temporal_attn_kernel<BF>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, w.q2, w.kv2,
                       w.o16, Mp, T, heads);
  }
  mark(st, nm[4]);
  }
  ALoadBlocked<BF> ao{w.o16, Mp, M, 0, 1, 1};
  launch_gemm<BF>(ao, p.wp, EpiRowsF32{out, res, p.bp, rm, C, 1.f}, M, C, Cp, st);
  mark(st, nm[5]);
  return AXVS_OK;
}

// A 128-row FFN tile takes 33 us where a 64-row tile takes 18 (one workgroup per CU; measured stand-alone, tools/ffn_wide_check.py):
// the wide kernel pays when its rounds of the 256-CU chip are so much fewer (21504 rows: 1 x 33 against 2 x 18).
bool ffn_wide_pays(long long M) {
  const long long r64 = ((M + kRows - 1) / kRows + 255) / 256, r128 = ((M + kWideRows - 1) / kWideRows + 255) / 256;
  return r64 > 1 && r128 * 11 < r64 * 6;
}

// norm1 -> linear1 -> ReLU -> linear2 -> +residual -> norm2 on fp32 rows X[M][C] (X is clobbered by the generic path)
template <bool BF>
int run_ffn(float* X, float* out, const LayerPacked& p, long long M, int C, int heads, int F, float* tmp, u16* y16, u16* h16,
            hipStream_t st, float* part = nullptr /* [F/256][M][256] fp32: enables the chunk-per-workgroup form for few rows */,
            RowStride rs = RowStride{0, 0} /* X and out rows: frames of rs.hw rows, rs.hw + rs.extra rows apart (fused kernels only) */) {
  const int oflags = g_out_dtype ? (g_out_dtype == 1 ? kOutF16 : kOutBf16) : 0;
  if (oflags && !(ffn_kernel_is_fused(C, heads, F) && ffn_lds_bytes(F) <= 160 * 1024))
    return fail(AXVS_ERR_ARG, "layer_out_dtype: a 16-bit output map needs the fused FFN tier (C = 256, 8 heads, d_ffn a multiple of 256 up to 4096)");
  if (!oflags && part != nullptr && ffn_split_mode(C, heads, F, M) == 2) {
    const dim3 sgrid((unsigned)((M + kRows - 1) / kRows), F / 512);
    if (g_ffn_gelu) {
      if (int rc = ensure_max_lds(reinterpret_cast<const void*>(&ffn_split_kernel<BF, true, 2>))) return rc;
      hipLaunchKernelGGL((ffn_split_kernel<BF, true, 2>), sgrid, dim3(512), kFfnSplitLds, st, (const float*)X, p.w1, p.b1, p.w2, p.g1, p.be1, part, M, F, rs);
    } else {
      if (int rc = ensure_max_lds(reinterpret_cast<const void*>(&ffn_split_kernel<BF, false, 2>))) return rc;
      hipLaunchKernelGGL((ffn_split_kernel<BF, false, 2>), sgrid, dim3(512), kFfnSplitLds, st, (const float*)X, p.w1, p.b1, p.w2, p.g1, p.be1, part, M, F, rs);
    }
    hipLaunchKernelGGL(ffn_finish_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, (const float*)X, (const float*)part, p.b2, p.g1, p.be1,
                         p.g2, p.be2, out, M, F / 256, rs);
    mark(st, "norm1+ffn+norm2");
    return AXVS_OK;
  }
  if (!oflags && part != nullptr && ffn_split_mode(C, heads, F, M) == 1) {
    const dim3 sgrid((unsigned)((M + kRows - 1) / kRows), F / 256);
    if (g_ffn_gelu) {
      if (int rc = ensure_max_lds(reinterpret_cast<const void*>(&ffn_split_kernel<BF, true>))) return rc;
      hipLaunchKernelGGL((ffn_split_kernel<BF, true>), sgrid, dim3(512), kFfnSplitLds, st, (const float*)X, p.w1, p.b1, p.w2, p.g1, p.be1, part, M, F, rs);
    } else {
      if (int rc = ensure_max_lds(reinterpret_cast<const void*>(&ffn_split_kernel<BF>))) return rc;
      hipLaunchKernelGGL((ffn_split_kernel<BF>), sgrid, dim3(512), kFfnSplitLds, st, (const float*)X, p.w1, p.b1, p.w2, p.g1, p.be1, part, M, F, rs);
    }
    hipLaunchKernelGGL(ffn_finish_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, (const float*)X, (const float*)part, p.b2, p.g1, p.be1,
                         p.g2, p.be2, out, M, F / 256, rs);
    mark(st, "norm1+ffn+norm2");
    return AXVS_OK;
  }
  if (!oflags && ffn_kernel_is_fused(C, heads, F) && F <= 2048 && (g_ffn_wide == 1 || (g_ffn_wide == 0 && ffn_wide_pays(M)))) {
    // more 64-row tiles than CUs: 128-row tiles when that saves a round of the chip (axvs_ffn_wide.h; bit-identical)
    const size_t lds = ffn_wide_lds_bytes(F);
    const dim3 wgrid((unsigned)((M + kWideRows - 1) / kWideRows));
    if (g_ffn_gelu) {
      if (int rc = ensure_max_lds(reinterpret_cast<const void*>(&ffn_wide_kernel<BF, true>))) return rc;
      hipLaunchKernelGGL((ffn_wide_kernel<BF, true>), wgrid, dim3(512), lds, st, (const float*)X, p.w1, p.b1, p.w2, p.b2, p.g1, p.be1, p.g2, p.be2, out, M, F, rs);
    } else {
      if (int rc = ensure_max_lds(reinterpret_cast<const void*>(&ffn_wide_kernel<BF>))) return rc;
      hipLaunchKernelGGL((ffn_wide_kernel<BF>), wgrid, dim3(512), lds, st, (const float*)X, p.w1, p.b1, p.w2, p.b2, p.g1, p.be1, p.g2, p.be2, out, M, F, rs);
    }
    mark(st, "norm1+ffn+norm2");
    return AXVS_OK;
  }
  if (ffn_kernel_is_fused(C, heads, F)) {
    const size_t lds = ffn_lds_bytes(F);
    if (lds > 160 * 1024) return fail(AXVS_ERR_ARG, "d_ffn=%d too large for the fused FFN kernel", F);
    const dim3 fgrid((unsigned)((M + kRows - 1) / kRows));
    if (g_ffn_gelu) {
      if (int rc = ensure_max_lds(reinterpret_cast<const void*>(&ffn_fused_kernel<BF, true>))) return rc;
      hipLaunchKernelGGL((ffn_fused_kernel<BF, true>), fgrid, dim3(512), lds, st, X, p.w1, p.b1, p.w2, p.b2, p.g1, p.be1, p.g2, p.be2, out, M, F, rs, oflags);
    } else {
      if (int rc = ensure_max_lds(reinterpret_cast<const void*>(&ffn_fused_kernel<BF>))) return rc;
      hipLaunchKernelGGL((ffn_fused_kernel<BF>), fgrid, dim3(512), lds, st, X, p.w1, p.b1, p.w2, p.b2, p.g1, p.be1, p.g2, p.be2, out, M, F, rs, oflags);
    }
    mark(st, "norm1+ffn+norm2");
    return AXVS_OK;
  }
  if (rs.hw) return fail(AXVS_ERR_ARG, "internal: strided frames need the fused FFN kernels");
  const unsigned lnblocks = (unsigned)((M + 3) / 4);
  hipLaunchKernelGGL((layernorm_kernel<BF>), dim3(lnblocks), dim3(256), 0, st, X, p.g1, p.be1, tmp, y16, M, C, 1e-5f);
  mark(st, "norm1");
  ALoadBlocked<BF> ay{y16, M, (int)M, 0, 1, 1};
  launch_gemm<BF>(ay, p.w1, EpiBlocked16<BF>{h16, M, p.b1, 1.f, 0, g_ffn_gelu ? 2 : 1}, (int)M, F, C, st);
  mark(st, "ffn.linear1");
  ALoadBlocked<BF> ah{h16, M, (int)M, 0, 1, 1};
  launch_gemm<BF>(ah, p.w2, EpiRowsF32{X, tmp, p.b2, identity_map(M), C, 1.f}, (int)M, C, F, st);
  mark(st, "ffn.linear2");
  hipLaunchKernelGGL((layernorm_kernel<BF>), dim3(lnblocks), dim3(256), 0, st, X, p.g2, p.be2, out, (u16*)nullptr, M, C, 1e-5f);
  mark(st, "norm2");
  return AXVS_OK;
}

int last_launch_status() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(AXVS_ERR_LAUNCH, "HIP launch failed: %s", hipGetErrorString(e));
  return AXVS_OK;
}

template <bool BF>
int traj_attn_fwd_t(const float* query, const float* key, const float* value, float* out, float* attn, const void* packed,
                    int S, int T, int L, int C, int heads, void* ws, hipStream_t st) {
  Carver pc(const_cast<void*>(packed));
  TrajPacked p = carve_traj(pc, C, heads);
  Carver wc(ws);
  TrajWs w = carve_traj_ws(wc, (long long)S * T * L, T, heads, false, padded_rows((long long)S * T * L, L));
  // the fused kernels always add a residual: feed zeros here (TrajectoryAttention.forward itself has none)
  float* zeros = wc.take<float>((size_t)S * T * L * C);
  if (hipMemsetAsync(zeros, 0, (size_t)S * T * L * C * sizeof(float), st) != hipSuccess) return fail(AXVS_ERR_LAUNCH, "memset failed");
  RowMap rm{T * L, L, 1, (long long)T * L, L, 1, 0};
  int rc = run_traj<BF>(query, key, value, nullptr, zeros, out, attn, p, w, rm, S, T, L, C, heads, st);
  return rc != AXVS_OK ? rc : last_launch_status();
}

bool sine_in_kernel(int C, int heads) { return !g_generic_only && C == 256 && heads == 8; }

PosGen make_posgen(const AxvsSinePos3D& sp, int T, int H, int W, int C, int l_is_h) {
  PosGen pg{};
  pg.mode = 1;
  pg.l_is_h = l_is_h;
  const float eps = 1e-6f;
  pg.zs = sp.normalize ? sp.scale / ((float)T + eps) : 1.f;
  pg.ys = sp.normalize ? sp.scale / ((float)H + eps) : 1.f;
  pg.xs = sp.normalize ? sp.scale / ((float)W + eps) : 1.f;
  pg.n = C / 2;
  pg.ke_yx = -log2f(sp.temperature) * 2.f / (float)pg.n;
  pg.ke_z = -log2f(sp.temperature) * 2.f / (float)C;
  pg.level = sp.level_embed;
  return pg;
}

template <bool BF>
int axial_layer_fwd_t(const float* src, const float* pos, float* out, const void* packed, int B, int T, int H, int W, int C,
                      int heads, int F, void* ws, float* h_attn, float* w_attn, hipStream_t st, const AxvsSinePos3D* sine = nullptr,
                      int which = 0 /* 0: whole layer; 1: height pass only (out = src + height_attn); 2: width pass + norm1 + FFN + norm2 on src */,
                      long long fs = 0 /* > 0: frames of src / out (and of the row-addressed temporaries) are fs rows apart; out may be src */) {
  Carver pc(const_cast<void*>(packed));
  LayerPacked p = carve_layer(pc, C, heads, F);
  const long long M = (long long)B * T * H * W;
  const long long span = fs ? ((long long)B * T - 1) * fs + (long long)H * W : M;      // rows spanned by a row-addressed tensor
  struct SpanGuard { SpanGuard(long long v) { g_row_span = v; } ~SpanGuard() { g_row_span = 0; } } span_guard(fs ? span : 0);
  if (g_out_dtype && (fs != 0 || which == 1))
    return fail(AXVS_ERR_ARG, "layer_out_dtype: a 16-bit output map exists for the whole layer / its width pass on contiguous frames only");
  Carver wc(ws);
  const LayerPlan plan = plan_layer(B, T, H, W, C, heads, F, h_attn != nullptr || w_attn != nullptr);
  const long long Mq = std::max(padded_rows(M, H), padded_rows(M, W));      // q/k/v row space: frames padded to multiples of 16 rows
  TrajWs tw = carve_traj_ws(wc, M, T, heads, plan.lean_traj, Mq);
  float* buf1 = wc.take<float>((size_t)span * C);
  float* const scratch1 = buf1;                // fp32 scratch of the generic FFN path (free once the width pass has read the rows)
  float* buf2 = plan.need_buf2 ? wc.take<float>((size_t)span * C) : nullptr;
  u16* y16 = plan.need_ffn_tmp ? wc.take<u16>((size_t)M * C) : nullptr;
  u16* h16 = plan.need_ffn_tmp ? wc.take<u16>((size_t)M * F) : nullptr;
  float* ffn_part = plan.need_ffn_part ? wc.take<float>((size_t)(F / 256) * M * C) : nullptr;
  const long long sB = fs ? (long long)T * fs : (long long)T * H * W, sT = fs ? fs : (long long)H * W;

  g_prof_next = 0;
  mark(st, "begin");
  // positions given as a PositionEmbeddingSine3D specification: the fused QKV kernel evaluates them (no HBM read); the other
  // tiers get them materialised into the workspace first
  PosGen pgh{}, pgw{};
  const PosGen *ph = nullptr, *pw = nullptr;
  if (sine) {
    if (sine_in_kernel(C, heads)) {
      pgh = make_posgen(*sine, T, H, W, C, 1);
      pgw = make_posgen(*sine, T, H, W, C, 0);
      ph = &pgh;
      pw = &pgw;
    } else {
      float* pbuf = wc.take<float>((size_t)M * C);
      const long long total = (long long)T * H * W * C;
      hipLaunchKernelGGL(pos3d_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, pbuf, B, T, H, W, C, sine->temperature,
                         sine->normalize, sine->scale);
      if (sine->level_embed)
        hipLaunchKernelGGL(add_channel_vector_kernel, dim3((unsigned)(((size_t)M * C + 255) / 256)), dim3(256), 0, st, pbuf, sine->level_embed,
                           (size_t)M * C, C);
      pos = pbuf;
      mark(st, "pos3d");
    }
  }
  // height pass: sequences (b, w), tokens (t, h)        WC/temporal_attention.py:197-204
  RowMap rmh{T * H, H, W, sB, sT, W, 1};
  int rc = AXVS_OK;
  if (which != 2) {
    rc = run_traj<BF>(src, src, src, pos, src, which == 1 ? out : buf1, h_attn, p.th, tw, rmh, B * W, T, H, C, heads, st, 1, nullptr, nullptr, nullptr, ph, true);
    if (rc != AXVS_OK) return rc;
    if (which == 1) return last_launch_status();
  } else {
    buf1 = const_cast<float*>(src);            // the caller's tensor IS the height pass's output (read only below)
  }
  // width pass: sequences (b, h), tokens (t, w)         :206-213
  RowMap rmw{T * W, W, H, sB, sT, 1, W};
  const FfnArgs fa{p.w1, p.w2, p.b1, p.b2, p.g1, p.be1, p.g2, p.be2, F};
  bool ffn_done = false;
  rc = run_traj<BF>(buf1, buf1, buf1, pos, buf1, buf2, w_attn, p.tw, tw, rmw, B * H, T, W, C, heads, st, 2, &fa, out, &ffn_done, pw, true);
  if (rc != AXVS_OK) return rc;
  if (ffn_done) return last_launch_status();   // the width-pass kernel ran norm1 -> FFN -> norm2 too and wrote `out`

  // norm1 -> FFN -> norm2                               :181-185, :217-218
  int rc2 = run_ffn<BF>(buf2, out, p, M, C, heads, F, scratch1, y16, h16, st, ffn_part, fs ? RowStride{H * W, fs - (long long)H * W} : RowStride{0, 0});
  if (rc2 != AXVS_OK) return rc2;
  return last_launch_status();
}

// TemporalTrajectoryAttentionLayer (WC/temporal_attention.py:103-155): ONE trajectory attention over all T*H*W tokens of a clip
// (frames of H*W keys), then norm1 -> FFN -> norm2.  Packed blob: TrajPacked | FFN part of LayerPacked.
template <bool BF>
int traj_layer_fwd_t(const float* src, const float* pos, float* out, const void* packed, int B, int T, int HW, int C, int heads, int F,
                     void* ws, hipStream_t st) {
  Carver pc(const_cast<void*>(packed));
  TrajPacked pt = carve_traj(pc, C, heads);
  LayerPacked pf = carve_ffn(pc, C, F);
  const long long M = (long long)B * T * HW;
  Carver wc(ws);
  TrajWs tw = carve_traj_ws(wc, M, T, heads, false, padded_rows(M, HW));
  float* x = wc.take<float>((size_t)M * C);
  float* tmp = wc.take<float>((size_t)M * C);
  u16* y16 = wc.take<u16>((size_t)M * C);
  u16* h16 = wc.take<u16>((size_t)M * F);
  g_prof_next = 0;
  mark(st, "begin");
  // src [(B T), HW, C] is already the sequence order 'B (T HW) C': identity row map, T frames of HW keys
  RowMap rm{T * HW, HW, 1, (long long)T * HW, HW, 1, 0};
  int rc = run_traj<BF>(src, src, src, pos, src, x, nullptr, pt, tw, rm, B, T, HW, C, heads, st, 0);
  if (rc != AXVS_OK) return rc;
  rc = run_ffn<BF>(x, out, pf, M, C, heads, F, tmp, y16, h16, st);
  return rc != AXVS_OK ? rc : last_launch_status();
}

// ---------------- cross-clip module ----------------
struct CCLayerPacked {
  TrajPacked t;
  u16* aspp_taps;                      // the folded temporal ASPP (axvs_cc.h, pack_aspp_taps_kernel): blocked weights [K = 7 * 256][256]
  float *aspp_bias, *norm_w, *norm_b, *an_w, *an_b, *cn_w, *cn_b;
};
CCLayerPacked carve_cc_layer(Carver& c) {
  CCLayerPacked l;
  l.t = carve_traj(c, 256, 8);
  l.aspp_taps = c.take<u16>(7 * 256 * 256);
  l.aspp_bias = c.take<float>(256);
  l.norm_w = c.take<float>(256); l.norm_b = c.take<float>(256);
  l.an_w = c.take<float>(256); l.an_b = c.take<float>(256);
  l.cn_w = c.take<float>(256); l.cn_b = c.take<float>(256);
  return l;
}
struct CCHeadsPacked {
  u16 *wemb, *wmh;                       // [512,256] (class | mask projection), [128,256]
  float *emb_mul, *emb_add, *mh_mul, *mh_add, *wc, *bc, *wa, *ba, *pix;   // folded BN; class / activation heads fp32; pixel BN (mul, add)
};
CCHeadsPacked carve_cc_heads(Carver& c, int K1) {
  CCHeadsPacked h;
  h.wemb = c.take<u16>(512 * 256);
  h.wmh = c.take<u16>(128 * 256);
  h.emb_mul = c.take<float>(512); h.emb_add = c.take<float>(512);
  h.mh_mul = c.take<float>(128); h.mh_add = c.take<float>(128);
  h.wc = c.take<float>((size_t)K1 * 256); h.bc = c.take<float>(K1);
  h.wa = c.take<float>(256); h.ba = c.take<float>(4);
  h.pix = c.take<float>(4);
  return h;
}
struct CCLayerWs {
  TrajWs tw;
  float *t1, *t2, *y;
};
CCLayerWs carve_cc_layer_ws(Carver& c, long long R, int Tc, int Q) {
  CCLayerWs w;
  w.tw = carve_traj_ws(c, R, Tc, 8, false, padded_rows(R, Q));      // (Tube-Link: 100 queries per clip -> frames of 112 rows)
  w.t1 = c.take<float>((size_t)R * 256);
  w.t2 = c.take<float>((size_t)R * 256);
  w.y = c.take<float>((size_t)R * 256);
  return w;
}

void copy_f32(const float* src, float* dst, int n, hipStream_t st) {
  PackDim d{n, n, 0, 0, 0};
  pack_b(src, dst, d, st);
}
void fold_bn(const AxvsBN& bn, float* mul, float* add, int n, hipStream_t st) {
  hipLaunchKernelGGL(bn_fold_kernel, dim3((n + 255) / 256), dim3(256), 0, st, bn.w, bn.b, bn.mean, bn.var, 1e-3f, mul, add, n);
}

template <bool BF>
int cc_layer_fwd_t(const float* x, float* out, const void* packed, int B, int Q, int Tc, const int* rates, void* ws, hipStream_t st,
                   float* out2 = nullptr /* optional second copy of the output rows */) {
  Carver pc(const_cast<void*>(packed));
  CCLayerPacked p = carve_cc_layer(pc);
  const long long R = (long long)B * Q * Tc;
  Carver wc(ws);
  CCLayerWs w = carve_cc_layer_ws(wc, R, Tc, Q);
  g_prof_next = 0;
  mark(st, "begin");
  // trajectory attention over (t q) tokens of each video, read in place from [B,Q,Tc,C]:  row (b; t,q) -> b*Q*Tc + q*Tc + t
  RowMap rm{Tc * Q, Q, 1, (long long)Q * Tc, 1, Tc, 0};
  const unsigned lnblocks = (unsigned)((R + 3) / 4);
  // the post-norm LayerNorm(x + attn(x)) rides in the trajectory kernel's row-wise epilogue when a fused kernel runs
  const bool ln_in_kernel = !g_generic_only && fused_frames(Tc, R);
  if (ln_in_kernel) {
    p.t.post_ln_g = p.norm_w;
    p.t.post_ln_b = p.norm_b;
  }
  int rc = run_traj<BF>(x, x, x, nullptr, x, ln_in_kernel ? w.t2 : w.t1, nullptr, p.t, w.tw, rm, B, Tc, Q, 256, 8, st, 0, nullptr, nullptr, nullptr, nullptr, true);
  if (rc != AXVS_OK) return rc;
  if (!ln_in_kernel) {
    hipLaunchKernelGGL((layernorm_kernel<BF>), dim3(lnblocks), dim3(256), 0, st, w.t1, p.norm_w, p.norm_b, w.t2, (u16*)nullptr, R, 256,
                       1e-5f);
    mark(st, "cc.norm");
  }
  // temporal ASPP, folded at pack time into ONE linear map of the clip axis (three dilated 3-tap branches + concat + 1x1 projection:
  // CC/maxtron_cross_clip_tracking_module.py:176-201): y[t] = sum_j M_j x[clamp(t + off_j)] + b', 7 taps, K = 1792
  {
    ALoadTaps7<BF> at{w.t2, Tc, (int)R, {0, -rates[0], rates[0], -rates[1], rates[1], -rates[2], rates[2]}};
    launch_gemm<BF>(at, p.aspp_taps, EpiRowsF32{w.y, nullptr, p.aspp_bias, identity_map(R), 256, 1.f}, (int)R, 256, 7 * 256, st, 7);
  }
  mark(st, "cc.aspp");
  hipLaunchKernelGGL(cc_aspp_post_kernel, dim3(lnblocks), dim3(256), 0, st, w.y, w.t2, p.an_w, p.an_b, p.cn_w, p.cn_b, out, R, out2, g_cc_aspp_affine);
  mark(st, "cc.aspp_post");
  return last_launch_status();
}

// embeddings + mask-head kernels (-> kern16, blocked [4][R][32]) + class head; the mask einsum follows separately so that the
// module loop can run it once for all layers
// x: the clip queries of `nl` layers back to back ([nl][R][256]); the projections share their weights across layers, so all layers
// go through ONE launch per GEMM (kern16: blocked [4][nl*R][32], logits [nl][Q][K1])
template <bool BF>
int cc_heads_small_t(const float* x, float* logits, u16* kern16, const void* packed, int B, int Q, int Tc, int K1, float* emb, hipStream_t st,
                     int nl = 1) {
  Carver pc(const_cast<void*>(packed));
  CCHeadsPacked p = carve_cc_heads(pc, K1);
  const long long R = (long long)nl * B * Q * Tc;
  ALoadRowsLd<BF> ax{x, 256, 0, (int)R};
  EpiRowsF32 ee{emb, nullptr, p.emb_add, identity_map(R), 512, 1.f};
  ee.mul = p.emb_mul;
  ee.gelu = 1;
  launch_gemm<BF>(ax, p.wemb, ee, (int)R, 512, 256, st);
  ALoadRowsLd<BF> am{emb, 512, 256, (int)R};
  EpiBlocked16<BF> ek{kern16, R, p.mh_add, 1.f, 0, 0};
  ek.mul = p.mh_mul;
  launch_gemm<BF>(am, p.wmh, ek, (int)R, 128, 256, st);
  mark(st, "cc.embeddings");
  const float void_bias = logf((float)(K1 - 1) * 0.9f / (1.f - 0.9f));
  hipLaunchKernelGGL(cc_class_head_kernel, dim3(Q, nl), dim3(256), 0, st, emb, 512, p.wa, p.ba, p.wc, p.bc, logits, B, Q, Tc, K1, void_bias);
  mark(st, "cc.class_head");
  return AXVS_OK;
}

// masks of `nl` layers (kernels kstride apart, outputs ostride apart) from one pass over the pixel features
template <bool BF>
int cc_masks_t(const float* pf, const u16* kern16, float* masks, const void* packed, int B, int Q, int Tc, int V, int H, int W, int K1, int nl,
               long long kstride, long long ostride, hipStream_t st) {
  Carver pc(const_cast<void*>(packed));
  CCHeadsPacked p = carve_cc_heads(pc, K1);
  const long long R = (long long)nl * B * Q * Tc, P = (long long)V * H * W;      // rows of the blocked kernel matrix: all layers
  dim3 grid((unsigned)((P + kEinsumPx - 1) / kEinsumPx), B * Tc);
  const long long TP = (long long)Tc * P;
  const EinsumMap mp{128 * TP, P, TP, (long long)Q * TP, P, TP, Tc, 1};
  if (int rc = ensure_max_lds(reinterpret_cast<const void*>(&mask_einsum_kernel<BF, 128>))) return rc;
  // every stride of the map is a multiple of P: the pixel rows are as aligned as P (and the two base pointers) allow
  const int al = std::min(row_align(pf, P, P), row_align(masks, P, P));
  if (al == 4 && P % 4 == 0) {
    hipLaunchKernelGGL((mask_einsum_kernel<BF, 128>), grid, dim3(256), einsum_lds_bytes<128>(), st, pf, kern16, masks, Q, Tc, P, R, mp, p.pix, nl, kstride, ostride, al);
  } else {
    if (int rc = ensure_max_lds(reinterpret_cast<const void*>(&mask_einsum_kernel<BF, 128, true>))) return rc;
    hipLaunchKernelGGL((mask_einsum_kernel<BF, 128, true>), grid, dim3(256), einsum_lds_bytes<128>(), st, pf, kern16, masks, Q, Tc, P, R, mp, p.pix, nl, kstride, ostride, al);
  }
  mark(st, "cc.mask_einsum");
  return AXVS_OK;
}

template <bool BF>
int cc_heads_fwd_t(const float* x, const float* pf, float* logits, float* masks, const void* packed, int B, int Q, int Tc, int V, int H,
                   int W, int K1, void* ws, hipStream_t st) {
  const long long R = (long long)B * Q * Tc;
  Carver wc(ws);
  float* emb = wc.take<float>((size_t)R * 512);
  u16* kern16 = wc.take<u16>((size_t)R * 128);
  g_prof_next = 0;
  mark(st, "begin");
  cc_heads_small_t<BF>(x, logits, kern16, packed, B, Q, Tc, K1, emb, st);
  cc_masks_t<BF>(pf, kern16, masks, packed, B, Q, Tc, V, H, W, K1, 1, 0, 0, st);
  return last_launch_status();
}

// ---------------- Tube-Link cross-clip heads ----------------
struct TLHeadsPacked {
  u16 *w0, *w1, *w2;                                     // mask_embed MLP [256,256], [256,256], [Cm,256]
  float *b0, *b1, *b2, *pn_w, *pn_b, *wa, *ba, *wc, *bc;  // biases, post_norm, activation_proj, cls_embed (fp32)
};
TLHeadsPacked carve_tl_heads(Carver& c, int K1, int Cm) {
  TLHeadsPacked h;
  h.w0 = c.take<u16>(256 * 256); h.w1 = c.take<u16>(256 * 256); h.w2 = c.take<u16>((size_t)Cm * 256);
  h.b0 = c.take<float>(256); h.b1 = c.take<float>(256); h.b2 = c.take<float>(Cm);
  h.pn_w = c.take<float>(256); h.pn_b = c.take<float>(256);
  h.wa = c.take<float>(256); h.ba = c.take<float>(1);
  h.wc = c.take<float>((size_t)K1 * 256); h.bc = c.take<float>(K1);
  return h;
}

struct TLHeadsWs {
  float* xn;
  u16 *xn16, *h1, *h2;
};
TLHeadsWs carve_tl_heads_ws(Carver& c, long long R) {
  TLHeadsWs w;
  w.xn = c.take<float>((size_t)R * 256);
  w.xn16 = c.take<u16>((size_t)R * 256);
  w.h1 = c.take<u16>((size_t)R * 256);
  w.h2 = c.take<u16>((size_t)R * 256);
  return w;
}

template <bool BF>
int tl_heads_small_t(const float* x, float* logits, u16* kern16, const void* packed, int B, int Q, int Tc, int K1, int Cm, const TLHeadsWs& w,
                     hipStream_t st, int nl = 1) {
  Carver pc(const_cast<void*>(packed));
  TLHeadsPacked p = carve_tl_heads(pc, K1, Cm);
  const long long R = (long long)nl * B * Q * Tc;
  hipLaunchKernelGGL((layernorm_kernel<BF>), dim3((unsigned)((R + 3) / 4)), dim3(256), 0, st, x, p.pn_w, p.pn_b, w.xn, w.xn16, R, 256, 1e-5f);
  mark(st, "tl.post_norm");
  hipLaunchKernelGGL(tl_class_head_kernel, dim3((unsigned)(B * Q), nl), dim3(256), 0, st, w.xn, p.wa, p.ba, p.wc, p.bc, logits, Tc, K1);
  mark(st, "tl.class_head");
  launch_gemm<BF>(ALoadBlocked<BF>{w.xn16, R, (int)R, 0, 1, 1}, p.w0, EpiBlocked16<BF>{w.h1, R, p.b0, 1.f, 0, 1}, (int)R, 256, 256, st);
  launch_gemm<BF>(ALoadBlocked<BF>{w.h1, R, (int)R, 0, 1, 1}, p.w1, EpiBlocked16<BF>{w.h2, R, p.b1, 1.f, 0, 1}, (int)R, 256, 256, st);
  launch_gemm<BF>(ALoadBlocked<BF>{w.h2, R, (int)R, 0, 1, 1}, p.w2, EpiBlocked16<BF>{kern16, R, p.b2, 1.f, 0, 0}, (int)R, Cm, 256, st);
  mark(st, "tl.mask_embed");
  return AXVS_OK;
}

template <bool BF>
int tl_masks_t(const float* mf, const u16* kern16, float* masks, int B, int Q, int Tc, int fpc, int h, int w, int Cm, int nl, long long kstride,
               long long ostride, hipStream_t st) {
  const long long R = (long long)nl * B * Q * Tc, P = (long long)h * w;
  const int T = Tc * fpc;
  dim3 grid((unsigned)((P + kEinsumPx - 1) / kEinsumPx), B * T);
  const EinsumMap mp{(long long)T * Cm * P, (long long)Cm * P, P, (long long)T * Q * P, (long long)Q * P, P, T, fpc};
  const int al = std::min(row_align(mf, P, P), row_align(masks, P, P));
  const bool gen = !(al == 4 && P % 4 == 0);
#define AXVS_EINSUM(CK_, GEN_)                                                                                                            \
  {                                                                                                                                        \
    if (int rc = ensure_max_lds(reinterpret_cast<const void*>(&mask_einsum_kernel<BF, CK_, GEN_>))) return rc;                           \
    hipLaunchKernelGGL((mask_einsum_kernel<BF, CK_, GEN_>), grid, dim3(256), einsum_lds_bytes<CK_>(), st, mf, kern16, masks, Q, Tc, P, R, mp, \
                       (const float*)nullptr, nl, kstride, ostride, al);                                                                   \
  }
  if (Cm == 128) {
    if (gen) AXVS_EINSUM(128, true) else AXVS_EINSUM(128, false)
  } else {
    if (gen) AXVS_EINSUM(256, true) else AXVS_EINSUM(256, false)
  }
#undef AXVS_EINSUM
  mark(st, "tl.mask_einsum");
  return AXVS_OK;
}

template <bool BF>
int tl_heads_fwd_t(const float* x, const float* mf, float* logits, float* masks, const void* packed, int B, int Q, int Tc, int fpc,
                   int h, int w, int K1, int Cm, void* ws, hipStream_t st) {
  const long long R = (long long)B * Q * Tc;
  Carver wc(ws);
  const TLHeadsWs hw = carve_tl_heads_ws(wc, R);
  u16* kern16 = wc.take<u16>((size_t)R * Cm);
  g_prof_next = 0;
  mark(st, "begin");
  tl_heads_small_t<BF>(x, logits, kern16, packed, B, Q, Tc, K1, Cm, hw, st);
  tl_masks_t<BF>(mf, kern16, masks, B, Q, Tc, fpc, h, w, Cm, 1, 0, 0, st);
  return last_launch_status();
}

// ---------------- multi-scale deformable attention ----------------
struct MsdaPacked {
  u16 *wv, *wq, *wo;            // value_proj [Cp,C] (head blocks), sampling_offsets|attention_weights [3MLP,C], output_proj [C,Cp]
  float *bv, *bq, *bo;
  float* wq32;                  // sampling_offsets | attention_weights as fp32 rows [3MLP][C]: operand of the 128 x 128 split-precision GEMM
  float *wv32, *wo32;           // value_proj / output_proj as fp32 rows [C][C] (head_dim 32: the head-block order is the natural one)
};
MsdaPacked carve_msda(Carver& c, int C, int heads, int L, int P) {
  const size_t Cp = (size_t)heads * 32, nq = (size_t)3 * heads * L * P;
  MsdaPacked m;
  m.wv = c.take<u16>(3 * Cp * C);          // split precision: (hi | lo | hi) along K
  m.wq = c.take<u16>(3 * ((nq + 15) & ~(size_t)15) * C);      // (weight rows are stored in groups of 16: wblk_off)
  m.wo = c.take<u16>(3 * (size_t)C * Cp);
  m.bv = c.take<float>(Cp);
  m.bq = c.take<float>(nq);
  m.bo = c.take<float>(C);
  m.wq32 = c.take<float>(nq * C);
  m.wv32 = c.take<float>((size_t)C * C);
  m.wo32 = c.take<float>((size_t)C * C);
  return m;
}

int msda_levels(const int* shapes, int L, int S, MsdaLevels* lv) {
  if (L <= 0 || L > kMsdaMaxLevels) return fail(AXVS_ERR_ARG, "n_levels=%d must be in 1..%d", L, kMsdaMaxLevels);
  long long start = 0;
  lv->L = L;
  for (int l = 0; l < L; ++l) {
    if (shapes[2 * l] <= 0 || shapes[2 * l + 1] <= 0) return fail(AXVS_ERR_ARG, "empty level %d", l);
    lv->H[l] = shapes[2 * l];
    lv->W[l] = shapes[2 * l + 1];
    lv->start[l] = (int)start;
    start += (long long)shapes[2 * l] * shapes[2 * l + 1];
  }
  if (start != S) return fail(AXVS_ERR_ARG, "spatial shapes cover %lld tokens, input has %d", start, S);   // modules/ms_deform_attn.py:96
  return AXVS_OK;
}

// Y[M][N] = epilogue(X[M][K] (+ X2) . W[N][K]^T) on the 128 x 128 split-precision kernel (axvs_gemm_nt.h), option msda_gemm = pieces
int launch_nt128(const float* X, const float* X2, const float* W, float* Y, long long M, int N, int K, const tr::GemmEpi& e, hipStream_t st,
                 bool feeds_residual = false, long long lda = 0 /* row stride of X in floats (0: K) */,
                 int zsplit = 1 /* > 1: split-K -- workgroup z writes the partial product of its k-steps to Y + z M N (no epilogue terms) */) {
  tr::GemmLd ld{lda ? lda : K, K, N, 0, X2};
  if (zsplit > 1) ld.ksteps = ((K + tr::kGK - 1) / tr::kGK + zsplit - 1) / zsplit;
  const dim3 grid((unsigned)((M + tr::kGT - 1) / tr::kGT), (unsigned)((N + tr::kGT - 1) / tr::kGT), (unsigned)(zsplit > 1 ? zsplit : 1));
  const bool exact = g_msda_gemm == 3 || (g_msda_gemm == 4 && feeds_residual), gen = tr::gemm_nt_general(ld, K), add = X2 != nullptr;
#define AXVS_NT128(NS_, GEN_, ADD_)                                                                                                  \
  {                                                                                                                                   \
    if (int rc = ensure_max_lds(reinterpret_cast<const void*>(tr::tr_gemm_nt_kernel<NS_, 1, GEN_, ADD_>))) return rc;                \
    hipLaunchKernelGGL((tr::tr_gemm_nt_kernel<NS_, 1, GEN_, ADD_>), grid, dim3(512), tr::gemm_nt_lds<NS_>(), st, X, W, Y, M, N, K, ld, e); \
  }
  if (gen) {
    if (exact) AXVS_NT128(3, true, false) else AXVS_NT128(2, true, false)
  } else if (add) {
    if (exact) AXVS_NT128(3, false, true) else AXVS_NT128(2, false, true)
  } else {
    if (exact) AXVS_NT128(3, false, false) else AXVS_NT128(2, false, false)
  }
#undef AXVS_NT128
  return AXVS_OK;
}
// rows from which the 128 x 128 kernel beats the 64 x 64 one (fewer rows: too few workgroups)
inline bool use_nt128(long long rows, int C, int heads) { return g_msda_gemm && rows >= 2048 && C % 4 == 0 && C / heads == 32; }

// what: 0 = whole module (value_proj, offsets | logits, gather, output_proj), 1 = up to the gather with fp32 rows out
// (`out` = sampled [N*Lq][C], no output_proj)
template <bool BF>
int msda_fwd_t(const float* query, const float* refp, int ref_dim, const float* input, const unsigned char* mask, const MsdaLevels& lv,
               float* out, const MsdaPacked& p, int N, int Lq, int S, int C, int heads, int P, void* ws, hipStream_t st,
               const float* qadd = nullptr, const float* residual = nullptr, int what = 0) {
  const int L = lv.L, Cp = heads * 32, nq = 3 * heads * L * P;
  const long long Rv = (long long)N * S, Rq = (long long)N * Lq;
  Carver wc(ws);
  u16* value16 = wc.take<u16>((size_t)Rv * Cp);
  float* qproj = wc.take<float>((size_t)Rq * nq);
  u16* o16 = wc.take<u16>(2 * (size_t)Rq * Cp);
  g_prof_next = 0;
  mark(st, "begin");
  // value_proj and output_proj feed the module output directly (no residual / norm inside the module): split precision
  const tr::Drop nodrop{0u, 0u, 0u, 1.f};
  if (use_nt128(Rv, C, heads)) {          // fp32 rows in, one 16-bit piece out in the blocked layout the gather reads
    tr::GemmEpi e{p.bv, 1.f, 0, nodrop, 0.f};
    e.out16 = value16;
    e.kind16 = BF ? 2 : 1;
    e.zero_rows = mask;
    if (int rc = launch_nt128(input, nullptr, p.wv32, nullptr, Rv, Cp, C, e, st)) return rc;
  } else {
    EpiBlocked16<BF> ev{value16, Rv, p.bv, 1.f, 0, 0};
    ev.zero_rows = mask;
    launch_gemm<BF>(ALoadRowsF32Split3<BF>{input, (int)Rv, C}, p.wv, ev, (int)Rv, Cp, 3 * C, st);
  }
  mark(st, "msda.value_proj");
  // sampling offsets | attention logits: fp32 rows in, fp32 rows out, [N Lq] x [3 heads L P] x C -- at a few thousand rows and more
  // the 128 x 128 split-precision kernel of the training tier (axvs_gemm_nt.h: fp32 operands split into bf16 pieces in its
  // loader) beats the 64 x 64 one (config 3, 21504 rows: the three projections of a deformable layer 27 + 59 + 30 us -> ~65 us)
  if (g_msda_gemm && Rq >= 2048 && C % 4 == 0 && nq % 4 == 0) {
    const tr::GemmEpi e{p.bq, 1.f, 0, nodrop, 0.f};
    if (int rc = launch_nt128(query, qadd, p.wq32, qproj, Rq, nq, C, e, st)) return rc;
  } else {
    launch_gemm<BF>(ALoadRowsF32Split3<BF>{query, (int)Rq, C, qadd}, p.wq, EpiRowsF32{qproj, nullptr, p.bq, identity_map(Rq), nq, 1.f}, (int)Rq,
                    nq, 3 * C, st);
  }
  mark(st, "msda.offsets+weights");
  const long long groups = Rq * heads;
  const dim3 ggrid((unsigned)((groups + 63) / 64));
  // the output projection on the 128 x 128 kernel reads the sampled rows as fp32 [N Lq][C] (same bytes as the two 16-bit pieces)
  const bool out128 = what != 1 && use_nt128(Rq, C, heads);
  float* of32 = what == 1 ? out : (out128 ? reinterpret_cast<float*>(o16) : nullptr);
  if (P == 4) hipLaunchKernelGGL((msda_gather_kernel<BF, 4>), ggrid, dim3(256), 0, st, value16, qproj, refp, ref_dim, lv, o16, N, S, Lq, heads, P, of32, C / heads);
  else hipLaunchKernelGGL((msda_gather_kernel<BF, 0>), ggrid, dim3(256), 0, st, value16, qproj, refp, ref_dim, lv, o16, N, S, Lq, heads, P, of32, C / heads);
  mark(st, "msda.gather");
  if (what == 1) return last_launch_status();
  if (out128) {
    tr::GemmEpi e{p.bo, 1.f, 0, nodrop, 0.f};
    e.res = residual;
    if (int rc = launch_nt128(of32, nullptr, p.wo32, out, Rq, C, C, e, st, true)) return rc;
  } else {
    launch_gemm<BF>(ALoadBlockedSplit3<BF>{o16, Rq, (int)Rq, Cp}, p.wo, EpiRowsF32{out, residual, p.bo, identity_map(Rq), C, 1.f}, (int)Rq,
                    C, 3 * Cp, st);
  }
  mark(st, "msda.output_proj");
  return last_launch_status();
}

}  // namespace

namespace {
struct ModuleWs {
  void *chain, *heads;
  float* q;            // [layers][R][256] clip queries after every layer: the predictor heads run on all of them at once
  u16* kern;           // blocked [Cm/32][layers*R][32] mask kernels of every layer
};
ModuleWs carve_module_ws(Carver& c, size_t chain_bytes, size_t heads_bytes, long long R, int layers, int Cm) {
  ModuleWs m;
  m.chain = c.take<char>(chain_bytes);
  m.heads = c.take<char>(heads_bytes);
  m.q = c.take<float>((size_t)layers * R * 256);
  m.kern = c.take<u16>((size_t)layers * R * Cm);
  return m;
}

// The layer chain (trajectory attention -> ASPP -> norms) of layer i+1 only needs layer i's clip queries, not its predictions, and
// the predictor heads share their weights across layers: the chain runs first, then heads(stream) computes the class logits and
// mask kernels of ALL layers (one launch per GEMM over layers*R rows) and the mask einsum of all layers in one pass over the pixel
// features (read once instead of once per layer).
template <class Heads>
int run_cc_module(const float* clip_query, const void* const* packed_layers, int layers, float* last_query, int B, int Q, int Tc, const int* rates,
                  int dtype, const ModuleWs& w, hipStream_t st, Heads heads) {
  const long long R = (long long)B * Q * Tc;
  if (layers <= 0 || layers > 64) return fail(AXVS_ERR_ARG, "num_layers=%d must be in 1..64", layers);
  const float* cur = clip_query;
  for (int i = 0; i < layers; ++i) {
    float* nxt = w.q + (size_t)i * R * 256;
    float* also = i == layers - 1 ? last_query : nullptr;      // the caller's copy of the last layer's queries
    int rc = dtype == AXVS_BF16 ? cc_layer_fwd_t<kBF>(cur, nxt, packed_layers[i], B, Q, Tc, rates, w.chain, st, also)
                                : cc_layer_fwd_t<false>(cur, nxt, packed_layers[i], B, Q, Tc, rates, w.chain, st, also);
    if (rc != AXVS_OK) return rc;
    cur = nxt;
  }
  if (int rc = heads(st)) return rc;
  return last_launch_status();
}
}  // namespace

// =====================================================================================
extern "C" {

int axvs_version(void) { return 1; }
int axvs_has_bf16(void) { return kBF ? 1 : 0; }

int axvs_profile_stages(void** events, int capacity) {
  g_prof_events = reinterpret_cast<hipEvent_t*>(events);
  g_prof_cap = events ? capacity : 0;
  return kMaxStages;
}
int axvs_profile_stage_count(void) { return g_prof_next < kMaxStages ? g_prof_next : kMaxStages; }
const char* axvs_profile_stage_name(int i) { return (i >= 0 && i < kMaxStages && g_stage_names[i]) ? g_stage_names[i] : ""; }

#if defined(AXVS_STAMPS) && defined(AXVS_STAMPS_QKV)   // the QKV kernel lives in this unit; the trajectory kernels: axvs_temporal_inst.hip
int axvs_debug_read_stamps(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(axvs::g_stamps), sizeof(unsigned long long) * n);
}
#endif

// 15 keys.  Functional switches the host modules set around their calls: ffn_gelu, layer_out_dtype, cc_aspp_affine, cc_last_heads_only, train_amp, no_merge_qkv
// (the 'verify' hand-off policy's re-run).  Tier selection for parity tests and the bench's QK^T/AV probe: generic_only, no_attn_fusion, no_ffn_fusion,
// merge_qkv_any, spatial_only, train_valu, train_exact.  Test hooks: sync_spin_limit, and plan_force -- ONE diagnostic bit mask that forces the planner's
// size-dependent choices for the bit-identity tests (forms that are bit-identical by construction, so the row count may decide):
//   1 never the 16-row trajectory tiles | 2 / 4 the 128-row FFN tiles always / never | 8 no two-chunk FFN workgroups | 16 / 32 merged launch on 16-row tiles
//   never / at any size | 64 generic tier without the reassociated temporal half.
// (Rounds 2 - 5 exposed 37 keys, most of them thresholds and forms measured slower; those are constants / gone since round 6: DESIGN.md.)
int axvs_set_option(const char* key, int value) {
  if (!key) return fail(AXVS_ERR_ARG, "null option key");
  if (!strcmp(key, "generic_only")) { g_generic_only = value; return AXVS_OK; }
  if (!strcmp(key, "no_attn_fusion")) { g_no_attn_fusion = value; return AXVS_OK; }
  if (!strcmp(key, "no_ffn_fusion")) { g_no_ffn_fusion = value; return AXVS_OK; }
  if (!strcmp(key, "ffn_gelu")) { g_ffn_gelu = value; return AXVS_OK; }
  if (!strcmp(key, "train_valu")) { g_train_valu = value; return AXVS_OK; }
  if (!strcmp(key, "train_exact")) { g_train_exact = value; return AXVS_OK; }
  if (!strcmp(key, "train_amp")) {
    if (value < 0 || value > 2) return fail(AXVS_ERR_ARG, "train_amp: 0 (off), 1 (bf16 products) or 2 (fp16 products)");
    g_train_amp = value;
    return AXVS_OK;
  }
  if (!strcmp(key, "spatial_only")) { g_spatial_only = value; return AXVS_OK; }
  if (!strcmp(key, "cc_aspp_affine")) { g_cc_aspp_affine = value ? 1 : 0; return AXVS_OK; }
  if (!strcmp(key, "cc_last_heads_only")) { g_cc_last_only = value; return AXVS_OK; }
  if (!strcmp(key, "no_merge_qkv")) { g_no_merge_qkv = value; return AXVS_OK; }
  if (!strcmp(key, "merge_qkv_any")) { g_merge_qkv_any = value; return AXVS_OK; }
  if (!strcmp(key, "layer_out_dtype")) {
    if (value < 0 || value > 2) return fail(AXVS_ERR_ARG, "layer_out_dtype: 0 (fp32), 1 (f16) or 2 (bf16)");
    g_out_dtype = value;
    return AXVS_OK;
  }
  if (!strcmp(key, "sync_spin_limit")) { g_sync_spin_limit = value > 0 ? (unsigned)value : axvs::kSyncSpinLimit; return AXVS_OK; }
  if (!strcmp(key, "plan_force")) {
    g_no_small_tiles = (value & 1) ? 1 : 0;
    g_ffn_wide = (value & 2) ? 1 : (value & 4) ? 2 : 0;
    g_ffn_split_pairs = (value & 8) ? 0 : 1;
    g_merge_small = (value & 16) ? 0 : (value & 32) ? 1 : kMergeSmall;
    g_no_reassoc = (value & 64) ? 1 : 0;
    return AXVS_OK;
  }
  return fail(AXVS_ERR_ARG, "unknown option");
}
const char* axvs_last_error(void) { return g_err; }

int axvs_set_status_buffer(int* device_word) {
  g_status = device_word;
  g_status_host = nullptr;
  if (device_word != nullptr) {
    // a word in pinned host memory (hipHostMalloc: device-visible at the same address) can be read by the host without a copy
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, device_word) == hipSuccess && at.type == hipMemoryTypeHost && at.hostPointer != nullptr)
      g_status_host = static_cast<volatile int*>(at.hostPointer);
    else
      (void)hipGetLastError();      // (an unregistered pointer leaves an error behind: not ours to report)
  }
  return AXVS_OK;
}

int axvs_check_status(void) { return status_gate(); }

int axvs_set_sync_buffer(unsigned* device_words, size_t n_words) {
  if (device_words != nullptr && n_words == 0) return fail(AXVS_ERR_ARG, "empty sync buffer");
  g_sync = device_words;
  g_sync_words = device_words ? n_words : 0;
  return AXVS_OK;
}

size_t axvs_traj_packed_bytes(int C, int heads) {
  Carver c(nullptr);
  carve_traj(c, C, heads);
  return c.off;
}

int axvs_traj_pack(const AxvsTrajParams* p, void* packed, int C, int heads, int dtype, void* stream) {
  if (int rcd = check_dtype(dtype)) return rcd;
  if (!p || !packed) return fail(AXVS_ERR_ARG, "null pointer");
  if (int rc = check_cfg(C, heads)) return rc;
  Carver c(packed);
  TrajPacked t = carve_traj(c, C, heads);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == AXVS_BF16) pack_traj<kBF>(*p, t, C, heads, st);
  else if (dtype == AXVS_F16) pack_traj<false>(*p, t, C, heads, st);
  else return fail(AXVS_ERR_ARG, "unknown dtype %d", dtype);
  return last_launch_status();
}

size_t axvs_axial_layer_packed_bytes(int C, int heads, int d_ffn) {
  Carver c(nullptr);
  carve_layer(c, C, heads, d_ffn);
  return c.off;
}

int axvs_axial_layer_pack(const AxvsAxialLayerParams* p, void* packed, int C, int heads, int d_ffn, int dtype, void* stream) {
  if (int rcd = check_dtype(dtype)) return rcd;
  if (!p || !packed) return fail(AXVS_ERR_ARG, "null pointer");
  if (int rc = check_cfg(C, heads)) return rc;
  if (d_ffn <= 0 || d_ffn % 32 != 0) return fail(AXVS_ERR_ARG, "d_ffn=%d must be a positive multiple of 32", d_ffn);
  if (int rcd = check_dtype(dtype)) return rcd;
  Carver c(packed);
  LayerPacked l = carve_layer(c, C, heads, d_ffn);
  hipStream_t st = static_cast<hipStream_t>(stream);
  PackDim plainC{C, C, 0, 0, 0}, plainF{d_ffn, d_ffn, 0, 0, 0};
  if (dtype == AXVS_BF16) {
    pack_traj<kBF>(p->height_attn, l.th, C, heads, st);
    pack_traj<kBF>(p->width_attn, l.tw, C, heads, st);
    pack_w<kBF>(p->linear1_w, l.w1, plainF, plainC, st);
    pack_w<kBF>(p->linear2_w, l.w2, plainC, plainF, st);
  } else {
    pack_traj<false>(p->height_attn, l.th, C, heads, st);
    pack_traj<false>(p->width_attn, l.tw, C, heads, st);
    pack_w<false>(p->linear1_w, l.w1, plainF, plainC, st);
    pack_w<false>(p->linear2_w, l.w2, plainC, plainF, st);
  }
  pack_b(p->linear1_b, l.b1, plainF, st);
  pack_b(p->linear2_b, l.b2, plainC, st);
  pack_b(p->norm1_w, l.g1, plainC, st);
  pack_b(p->norm1_b, l.be1, plainC, st);
  pack_b(p->norm2_w, l.g2, plainC, st);
  pack_b(p->norm2_b, l.be2, plainC, st);
  return last_launch_status();
}

size_t axvs_traj_attn_workspace_bytes(int S, int T, int L, int C, int heads) {
  Carver c(nullptr);
  carve_traj_ws(c, (long long)S * T * L, T, heads, false, padded_rows((long long)S * T * L, L));
  c.take<float>((size_t)S * T * L * C);
  return c.off;
}

int axvs_traj_attn_fwd(const float* query, const float* key, const float* value, float* out, float* space_attn,
                       const void* packed, int S, int T, int L, int C, int heads, int dtype, void* workspace,
                       size_t workspace_bytes, void* stream) {
  if (int rcd = check_dtype(dtype)) return rcd;
  if (!query || !key || !value || !out || !packed || !workspace) return fail(AXVS_ERR_ARG, "null pointer");
  if (S <= 0 || T <= 0 || L <= 0) return fail(AXVS_ERR_ARG, "empty shape S=%d T=%d L=%d", S, T, L);
  if (int rc = check_cfg(C, heads)) return rc;
  if (workspace_bytes < axvs_traj_attn_workspace_bytes(S, T, L, C, heads))
    return fail(AXVS_ERR_WORKSPACE, "workspace too small: %zu < %zu", workspace_bytes,
                axvs_traj_attn_workspace_bytes(S, T, L, C, heads));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == AXVS_BF16) return traj_attn_fwd_t<kBF>(query, key, value, out, space_attn, packed, S, T, L, C, heads, workspace, st);
  if (dtype == AXVS_F16) return traj_attn_fwd_t<false>(query, key, value, out, space_attn, packed, S, T, L, C, heads, workspace, st);
  return fail(AXVS_ERR_ARG, "unknown dtype %d", dtype);
}

// span: rows spanned by a row-addressed temporary (= M unless the frames are strided)
static size_t layer_ws_bytes(int B, int T, int H, int W, int C, int heads, int d_ffn, int want_attn_maps, int sine_pos, long long span) {
  const long long M = (long long)B * T * H * W;
  Carver c(nullptr);
  const LayerPlan plan = plan_layer(B, T, H, W, C, heads, d_ffn, want_attn_maps != 0);
  carve_traj_ws(c, M, T, heads, plan.lean_traj, std::max(padded_rows(M, H), padded_rows(M, W)));
  c.take<float>((size_t)span * C);
  if (plan.need_buf2) c.take<float>((size_t)span * C);
  if (plan.need_ffn_tmp) {
    c.take<u16>((size_t)M * C);
    c.take<u16>((size_t)M * d_ffn);
  }
  if (plan.need_ffn_part) c.take<float>((size_t)(d_ffn / 256) * M * C);
  if (sine_pos && !sine_in_kernel(C, heads)) c.take<float>((size_t)M * C);     // materialised positions (tiers without in-kernel evaluation)
  return c.off;
}
size_t axvs_axial_layer_workspace_bytes_ex(int B, int T, int H, int W, int C, int heads, int d_ffn, int want_attn_maps, int sine_pos) {
  return layer_ws_bytes(B, T, H, W, C, heads, d_ffn, want_attn_maps, sine_pos, (long long)B * T * H * W);
}

// frames of src / out `frame_stride_rows` rows apart (a level of the pixel decoder's concatenated token buffer, used in place): the
// row-addressed temporaries take the same stride, i.e. span ((B T - 1) stride + H W) rows each
size_t axvs_axial_layer_workspace_bytes_strided(int B, int T, int H, int W, int C, int heads, int d_ffn, long long frame_stride_rows) {
  return layer_ws_bytes(B, T, H, W, C, heads, d_ffn, 0, 1, ((long long)B * T - 1) * frame_stride_rows + (long long)H * W);
}

int axvs_axial_layer_strided_ok(int C, int heads, int d_ffn) {
  return !g_generic_only && sine_in_kernel(C, heads) && ffn_kernel_is_fused(C, heads, d_ffn) ? 1 : 0;
}

int axvs_axial_layer_fwd_sine3d_strided(const float* src, const AxvsSinePos3D* pos, float* out, const void* packed, int B, int T, int H, int W,
                                        int C, int heads, int d_ffn, int dtype, long long frame_stride_rows, void* workspace,
                                        size_t workspace_bytes, void* stream) {
  if (int rcd = check_dtype(dtype)) return rcd;
  if (!src || !pos || !out || !packed || !workspace) return fail(AXVS_ERR_ARG, "null pointer");
  if (B <= 0 || T <= 0 || H <= 0 || W <= 0) return fail(AXVS_ERR_ARG, "empty shape B=%d T=%d H=%d W=%d", B, T, H, W);
  if (int rc = check_cfg(C, heads)) return rc;
  if (d_ffn <= 0 || d_ffn % 32 != 0) return fail(AXVS_ERR_ARG, "d_ffn=%d must be a positive multiple of 32", d_ffn);
  if (!axvs_axial_layer_strided_ok(C, heads, d_ffn)) return fail(AXVS_ERR_ARG, "strided frames need the fused tier (C = 256, 8 heads, d_ffn a multiple of 256)");
  if (frame_stride_rows < (long long)H * W) return fail(AXVS_ERR_ARG, "frame stride %lld < H W = %d rows", frame_stride_rows, H * W);
  if (T > 255 || H > 4095 || W > 4095) return fail(AXVS_ERR_ARG, "grid too large for generated positions");
  const long long span = ((long long)B * T - 1) * frame_stride_rows + (long long)H * W;
  if (span > 2147483647LL / 64) return fail(AXVS_ERR_ARG, "too many rows for 32-bit row indices");
  if (int rcd = check_dtype(dtype)) return rcd;
  const size_t need = axvs_axial_layer_workspace_bytes_strided(B, T, H, W, C, heads, d_ffn, frame_stride_rows);
  if (workspace_bytes < need) return fail(AXVS_ERR_WORKSPACE, "workspace too small: %zu < %zu", workspace_bytes, need);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == AXVS_BF16)
    return axial_layer_fwd_t<kBF>(src, nullptr, out, packed, B, T, H, W, C, heads, d_ffn, workspace, nullptr, nullptr, st, pos, 0, frame_stride_rows);
  return axial_layer_fwd_t<false>(src, nullptr, out, packed, B, T, H, W, C, heads, d_ffn, workspace, nullptr, nullptr, st, pos, 0, frame_stride_rows);
}

size_t axvs_axial_layer_workspace_bytes(int B, int T, int H, int W, int C, int heads, int d_ffn) {
  return axvs_axial_layer_workspace_bytes_ex(B, T, H, W, C, heads, d_ffn, 1 /* upper bound: with attention maps */, 0);
}

static int axial_layer_entry(const float* src, const float* pos, const AxvsSinePos3D* sine, float* out, const void* packed, int B, int T,
                             int H, int W, int C, int heads, int d_ffn, int dtype, void* workspace, size_t workspace_bytes, float* h_attn,
                             float* w_attn, void* stream) {
  if (!src || (!pos && !sine) || !out || !packed || !workspace) return fail(AXVS_ERR_ARG, "null pointer");
  if (B <= 0 || T <= 0 || H <= 0 || W <= 0) return fail(AXVS_ERR_ARG, "empty shape B=%d T=%d H=%d W=%d", B, T, H, W);
  if (src == out) return fail(AXVS_ERR_ARG, "out may not alias src");
  if (int rc = check_cfg(C, heads)) return rc;
  if (d_ffn <= 0 || d_ffn % 32 != 0) return fail(AXVS_ERR_ARG, "d_ffn=%d must be a positive multiple of 32", d_ffn);
  if (sine && (T > 255 || H > 4095 || W > 4095)) return fail(AXVS_ERR_ARG, "grid too large for generated positions");
  const size_t need = axvs_axial_layer_workspace_bytes_ex(B, T, H, W, C, heads, d_ffn, h_attn != nullptr || w_attn != nullptr, sine != nullptr);
  if (workspace_bytes < need) return fail(AXVS_ERR_WORKSPACE, "workspace too small: %zu < %zu", workspace_bytes, need);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == AXVS_BF16)
    return axial_layer_fwd_t<kBF>(src, pos, out, packed, B, T, H, W, C, heads, d_ffn, workspace, h_attn, w_attn, st, sine);
  if (dtype == AXVS_F16)
    return axial_layer_fwd_t<false>(src, pos, out, packed, B, T, H, W, C, heads, d_ffn, workspace, h_attn, w_attn, st, sine);
  return fail(AXVS_ERR_ARG, "unknown dtype %d", dtype);
}

int axvs_axial_layer_fwd(const float* src, const float* pos, float* out, const void* packed, int B, int T, int H, int W,
                         int C, int heads, int d_ffn, int dtype, void* workspace, size_t workspace_bytes, float* h_attn,
                         float* w_attn, void* stream) {
  if (int rcd = check_dtype(dtype)) return rcd;
  if (!pos) return fail(AXVS_ERR_ARG, "null pointer");
  return axial_layer_entry(src, pos, nullptr, out, packed, B, T, H, W, C, heads, d_ffn, dtype, workspace, workspace_bytes, h_attn, w_attn, stream);
}

int axvs_axial_pass_fwd(const float* src, const float* pos, float* out, const void* packed, int pass, int B, int T, int H, int W, int C,
                        int heads, int d_ffn, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
  if (int rcd = check_dtype(dtype)) return rcd;
  if (!src || !pos || !out || !packed || !workspace) return fail(AXVS_ERR_ARG, "null pointer");
  if (pass != 0 && pass != 1) return fail(AXVS_ERR_ARG, "pass must be 0 (height) or 1 (width + FFN)");
  if (B <= 0 || T <= 0 || H <= 0 || W <= 0) return fail(AXVS_ERR_ARG, "empty shape B=%d T=%d H=%d W=%d", B, T, H, W);
  if (src == out) return fail(AXVS_ERR_ARG, "out may not alias src");
  if (int rc = check_cfg(C, heads)) return rc;
  if (d_ffn <= 0 || d_ffn % 32 != 0) return fail(AXVS_ERR_ARG, "d_ffn=%d must be a positive multiple of 32", d_ffn);
  if (workspace_bytes < axvs_axial_layer_workspace_bytes_ex(B, T, H, W, C, heads, d_ffn, 0, 0)) return fail(AXVS_ERR_WORKSPACE, "workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == AXVS_BF16) return axial_layer_fwd_t<kBF>(src, pos, out, packed, B, T, H, W, C, heads, d_ffn, workspace, nullptr, nullptr, st, nullptr, pass + 1);
  if (dtype == AXVS_F16) return axial_layer_fwd_t<false>(src, pos, out, packed, B, T, H, W, C, heads, d_ffn, workspace, nullptr, nullptr, st, nullptr, pass + 1);
  return fail(AXVS_ERR_ARG, "unknown dtype %d", dtype);
}

size_t axvs_axial_layer_sine3d_workspace_bytes(int B, int T, int H, int W, int C, int heads, int d_ffn) {
  return axvs_axial_layer_workspace_bytes_ex(B, T, H, W, C, heads, d_ffn, 1, 1);
}

int axvs_axial_layer_fwd_sine3d(const float* src, const AxvsSinePos3D* pos, float* out, const void* packed, int B, int T, int H, int W,
                                int C, int heads, int d_ffn, int dtype, void* workspace, size_t workspace_bytes, float* h_attn,
                                float* w_attn, void* stream) {
  if (int rcd = check_dtype(dtype)) return rcd;
  if (!pos) return fail(AXVS_ERR_ARG, "null pointer");
  if (!(pos->temperature > 0.f)) return fail(AXVS_ERR_ARG, "temperature must be positive");
  return axial_layer_entry(src, nullptr, pos, out, packed, B, T, H, W, C, heads, d_ffn, dtype, workspace, workspace_bytes, h_attn, w_attn, stream);
}

size_t axvs_traj_layer_packed_bytes(int C, int heads, int d_ffn) {
  Carver c(nullptr);
  carve_traj(c, C, heads);
  carve_ffn(c, C, d_ffn);
  return c.off;
}

int axvs_traj_layer_pack(const AxvsTrajLayerParams* p, void* packed, int C, int heads, int d_ffn, int dtype, void* stream) {
  if (int rcd = check_dtype(dtype)) return rcd;
  if (!p || !packed) return fail(AXVS_ERR_ARG, "null pointer");
  if (int rc = check_cfg(C, heads)) return rc;
  if (d_ffn <= 0 || d_ffn % 32 != 0) return fail(AXVS_ERR_ARG, "d_ffn=%d must be a positive multiple of 32", d_ffn);
  if (int rcd = check_dtype(dtype)) return rcd;
  Carver c(packed);
  TrajPacked t = carve_traj(c, C, heads);
  LayerPacked l = carve_ffn(c, C, d_ffn);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == AXVS_BF16) {
    pack_traj<kBF>(p->temporal_attn, t, C, heads, st);
    pack_ffn<kBF>(p->norm1_w, p->norm1_b, p->linear1_w, p->linear1_b, p->linear2_w, p->linear2_b, p->norm2_w, p->norm2_b, l, C, d_ffn, st);
  } else {
    pack_traj<false>(p->temporal_attn, t, C, heads, st);
    pack_ffn<false>(p->norm1_w, p->norm1_b, p->linear1_w, p->linear1_b, p->linear2_w, p->linear2_b, p->norm2_w, p->norm2_b, l, C, d_ffn, st);
  }
  return last_launch_status();
}

size_t axvs_traj_layer_workspace_bytes(int B, int T, int HW, int C, int heads, int d_ffn) {
  const long long M = (long long)B * T * HW;
  Carver c(nullptr);
  carve_traj_ws(c, M, T, heads, false, padded_rows(M, HW));
  c.take<float>((size_t)M * C);
  c.take<float>((size_t)M * C);
  c.take<u16>((size_t)M * C);
  c.take<u16>((size_t)M * d_ffn);
  return c.off;
}

int axvs_traj_layer_fwd(const float* src, const float* pos, float* out, const void* packed, int B, int T, int HW, int C, int heads, int d_ffn,
                        int dtype, void* workspace, size_t workspace_bytes, void* stream) {
  if (int rcd = check_dtype(dtype)) return rcd;
  if (!src || !pos || !out || !packed || !workspace) return fail(AXVS_ERR_ARG, "null pointer");
  if (B <= 0 || T <= 0 || HW <= 0) return fail(AXVS_ERR_ARG, "empty shape B=%d T=%d HW=%d", B, T, HW);
  if (src == out) return fail(AXVS_ERR_ARG, "out may not alias src");
  if (int rc = check_cfg(C, heads)) return rc;
  if (d_ffn <= 0 || d_ffn % 32 != 0) return fail(AXVS_ERR_ARG, "d_ffn=%d must be a positive multiple of 32", d_ffn);
  if (workspace_bytes < axvs_traj_layer_workspace_bytes(B, T, HW, C, heads, d_ffn)) return fail(AXVS_ERR_WORKSPACE, "workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == AXVS_BF16) return traj_layer_fwd_t<kBF>(src, pos, out, packed, B, T, HW, C, heads, d_ffn, workspace, st);
  if (dtype == AXVS_F16) return traj_layer_fwd_t<false>(src, pos, out, packed, B, T, HW, C, heads, d_ffn, workspace, st);
  return fail(AXVS_ERR_ARG, "unknown dtype %d", dtype);
}

size_t axvs_ffn_workspace_bytes(long long M, int C, int d_ffn) {
  Carver c(nullptr);
  c.take<float>((size_t)M * C);
  c.take<float>((size_t)M * C);
  c.take<u16>((size_t)M * C);
  c.take<u16>((size_t)M * d_ffn);
  return c.off;
}

int axvs_ffn_fwd(const float* x, float* out, const void* packed_layer, long long M, int C, int heads, int d_ffn, int dtype,
                 void* workspace, size_t workspace_bytes, void* stream) {
  if (int rcd = check_dtype(dtype)) return rcd;
  if (!x || !out || !packed_layer || !workspace) return fail(AXVS_ERR_ARG, "null pointer");
  if (M <= 0) return fail(AXVS_ERR_ARG, "empty input");
  if (int rc = check_cfg(C, heads)) return rc;
  if (workspace_bytes < axvs_ffn_workspace_bytes(M, C, d_ffn)) return fail(AXVS_ERR_WORKSPACE, "workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  Carver pc(const_cast<void*>(packed_layer));
  LayerPacked p = carve_layer(pc, C, heads, d_ffn);
  Carver wc(workspace);
  float* xin = wc.take<float>((size_t)M * C);
  float* tmp = wc.take<float>((size_t)M * C);
  u16* y16 = wc.take<u16>((size_t)M * C);
  u16* h16 = wc.take<u16>((size_t)M * d_ffn);
  if (hipMemcpyAsync(xin, x, (size_t)M * C * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
    return fail(AXVS_ERR_LAUNCH, "copy failed");
  g_prof_next = 0;
  int rc = dtype == AXVS_BF16 ? run_ffn<kBF>(xin, out, p, M, C, heads, d_ffn, tmp, y16, h16, st)
                              : run_ffn<false>(xin, out, p, M, C, heads, d_ffn, tmp, y16, h16, st);
  return rc != AXVS_OK ? rc : last_launch_status();
}

size_t axvs_cc_layer_packed_bytes(void) {
  Carver c(nullptr);
  carve_cc_layer(c);
  return c.off;
}

int axvs_cc_layer_pack(const AxvsCCLayerParams* p, void* packed, int dtype, void* stream) {
  if (int rcd = check_dtype(dtype)) return rcd;
  if (!p || !packed) return fail(AXVS_ERR_ARG, "null pointer");
  if (int rcd = check_dtype(dtype)) return rcd;
  Carver c(packed);
  CCLayerPacked l = carve_cc_layer(c);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned cb = (7 * 256 * 256 + 255) / 256;
  if (dtype == AXVS_BF16) {
    pack_traj<kBF>(p->attn, l.t, 256, 8, st);
    hipLaunchKernelGGL((pack_aspp_taps_kernel<kBF>), dim3(cb), dim3(256), 0, st, p->aspp_w[0], p->aspp_w[1], p->aspp_w[2], p->aspp_proj_w, l.aspp_taps);
  } else {
    pack_traj<false>(p->attn, l.t, 256, 8, st);
    hipLaunchKernelGGL((pack_aspp_taps_kernel<false>), dim3(cb), dim3(256), 0, st, p->aspp_w[0], p->aspp_w[1], p->aspp_w[2], p->aspp_proj_w, l.aspp_taps);
  }
  hipLaunchKernelGGL(pack_aspp_bias_kernel, dim3(1), dim3(256), 0, st, p->aspp_b[0], p->aspp_b[1], p->aspp_b[2], p->aspp_proj_w, l.aspp_bias);
  copy_f32(p->norm_w, l.norm_w, 256, st); copy_f32(p->norm_b, l.norm_b, 256, st);
  copy_f32(p->aspp_norm_w, l.an_w, 256, st); copy_f32(p->aspp_norm_b, l.an_b, 256, st);
  copy_f32(p->conv_norm_w, l.cn_w, 256, st); copy_f32(p->conv_norm_b, l.cn_b, 256, st);
  return last_launch_status();
}

size_t axvs_cc_layer_workspace_bytes(int B, int Q, int Tc) {
  Carver c(nullptr);
  carve_cc_layer_ws(c, (long long)B * Q * Tc, Tc, Q);
  return c.off;
}

int axvs_cc_layer_fwd(const float* clip_query, float* out, const void* packed, int B, int Q, int Tc, const int* rates, int dtype,
                      void* workspace, size_t workspace_bytes, void* stream) {
  if (int rcd = check_dtype(dtype)) return rcd;
  if (!clip_query || !out || !packed || !rates || !workspace) return fail(AXVS_ERR_ARG, "null pointer");
  if (B <= 0 || Q <= 0 || Tc <= 0) return fail(AXVS_ERR_ARG, "empty shape");
  if (clip_query == out) return fail(AXVS_ERR_ARG, "out may not alias clip_query");
  if (workspace_bytes < axvs_cc_layer_workspace_bytes(B, Q, Tc)) return fail(AXVS_ERR_WORKSPACE, "workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == AXVS_BF16) return cc_layer_fwd_t<kBF>(clip_query, out, packed, B, Q, Tc, rates, workspace, st);
  if (dtype == AXVS_F16) return cc_layer_fwd_t<false>(clip_query, out, packed, B, Q, Tc, rates, workspace, st);
  return fail(AXVS_ERR_ARG, "unknown dtype %d", dtype);
}

size_t axvs_cc_heads_packed_bytes(int K1) {
  Carver c(nullptr);
  carve_cc_heads(c, K1);
  return c.off;
}

int axvs_cc_heads_pack(const AxvsCCHeadParams* p, void* packed, int K1, int dtype, void* stream) {
  if (int rcd = check_dtype(dtype)) return rcd;
  if (!p || !packed || K1 <= 0) return fail(AXVS_ERR_ARG, "bad argument");
  if (int rcd = check_dtype(dtype)) return rcd;
  Carver c(packed);
  CCHeadsPacked h = carve_cc_heads(c, K1);
  hipStream_t st = static_cast<hipStream_t>(stream);
  PackDim n256{256, 256, 0, 0, 0}, n128{128, 128, 0, 0, 0};
  if (dtype == AXVS_BF16) {
    pack_w<kBF>(p->class_proj_w, h.wemb, n256, n256, st, 0, 512);
    pack_w<kBF>(p->mask_proj_w, h.wemb, n256, n256, st, 256, 512);
    pack_w<kBF>(p->mask_head_w, h.wmh, n128, n256, st);
  } else {
    pack_w<false>(p->class_proj_w, h.wemb, n256, n256, st, 0, 512);
    pack_w<false>(p->mask_proj_w, h.wemb, n256, n256, st, 256, 512);
    pack_w<false>(p->mask_head_w, h.wmh, n128, n256, st);
  }
  fold_bn(p->class_proj_bn, h.emb_mul, h.emb_add, 256, st);
  fold_bn(p->mask_proj_bn, h.emb_mul + 256, h.emb_add + 256, 256, st);
  fold_bn(p->mask_head_bn, h.mh_mul, h.mh_add, 128, st);
  fold_bn(p->pixel_bn, h.pix, h.pix + 1, 1, st);
  hipLaunchKernelGGL(transpose_k1x256_kernel, dim3((unsigned)((K1 * 256 + 255) / 256)), dim3(256), 0, st, p->class_head_w, h.wc, K1);      // [256][K1]: class index on the lanes
  copy_f32(p->class_head_b, h.bc, K1, st);
  copy_f32(p->act_head_w, h.wa, 256, st);
  copy_f32(p->act_head_b, h.ba, 1, st);
  return last_launch_status();
}

size_t axvs_cc_heads_workspace_bytes(int B, int Q, int Tc) {
  Carver c(nullptr);
  const long long R = (long long)B * Q * Tc;
  c.take<float>((size_t)R * 512);
  c.take<u16>((size_t)R * 128);
  return c.off;
}

int axvs_cc_heads_fwd(const float* clip_query, const float* panoptic_features, float* pred_logits, float* pred_masks,
                      const void* packed, int B, int Q, int Tc, int V, int H, int W, int K1, int dtype, void* workspace,
                      size_t workspace_bytes, void* stream) {
  if (int rcd = check_dtype(dtype)) return rcd;
  if (!clip_query || !panoptic_features || !pred_logits || !pred_masks || !packed || !workspace) return fail(AXVS_ERR_ARG, "null pointer");
  if (B <= 0 || Q <= 0 || Tc <= 0 || V <= 0 || H <= 0 || W <= 0 || K1 <= 0) return fail(AXVS_ERR_ARG, "empty shape");
  if (B * Tc > 1024) return fail(AXVS_ERR_ARG, "B*Tc > 1024 is not supported by the class head");
  if (workspace_bytes < axvs_cc_heads_workspace_bytes(B, Q, Tc)) return fail(AXVS_ERR_WORKSPACE, "workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == AXVS_BF16) return cc_heads_fwd_t<kBF>(clip_query, panoptic_features, pred_logits, pred_masks, packed, B, Q, Tc, V, H, W, K1, workspace, st);
  if (dtype == AXVS_F16) return cc_heads_fwd_t<false>(clip_query, panoptic_features, pred_logits, pred_masks, packed, B, Q, Tc, V, H, W, K1, workspace, st);
  return fail(AXVS_ERR_ARG, "unknown dtype %d", dtype);
}

// ---- the whole layer loop of the cross-clip modules in ONE call (CC/...:283-318, TLCC:925-950): the Python host of round 1 made
//      2 library calls + 3 allocations per layer and became the bottleneck once the kernels were fused (494 us of host time per
//      forward against ~430 us of GPU time at BASELINE config 4).  The layer chain (trajectory attention -> ASPP -> norms) of layer
//      i+1 only needs layer i's clip queries, not its predictions: with an auxiliary stream the predictor heads of layer i run
//      beside the chain of layer i+1 (fork / join with events, capturable into a HIP graph).
size_t axvs_cc_module_workspace_bytes(int B, int Q, int Tc, int num_layers) {
  Carver c(nullptr);
  carve_module_ws(c, axvs_cc_layer_workspace_bytes(B, Q, Tc), (size_t)num_layers * B * Q * Tc * 512 * sizeof(float), (long long)B * Q * Tc, num_layers, 128);
  return c.off;
}

int axvs_cc_module_fwd(const float* clip_query, const float* panoptic_features, float* pred_logits, float* pred_masks, float* last_query,
                       const void* const* packed_layers, const void* packed_heads, int num_layers, int B, int Q, int Tc, int V, int H, int W,
                       int K1, const int* rates, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
  if (int rcd = check_dtype(dtype)) return rcd;
  if (!clip_query || !panoptic_features || !pred_logits || !pred_masks || !last_query || !packed_layers || !packed_heads || !rates || !workspace)
    return fail(AXVS_ERR_ARG, "null pointer");
  if (B <= 0 || Q <= 0 || Tc <= 0 || V <= 0 || H <= 0 || W <= 0 || K1 <= 0 || num_layers <= 0) return fail(AXVS_ERR_ARG, "empty shape");
  if (B * Tc > 1024) return fail(AXVS_ERR_ARG, "B*Tc > 1024 is not supported by the class head");
  if (int rcd = check_dtype(dtype)) return rcd;
  if (workspace_bytes < axvs_cc_module_workspace_bytes(B, Q, Tc, num_layers)) return fail(AXVS_ERR_WORKSPACE, "workspace too small");
  const long long R = (long long)B * Q * Tc;
  Carver wc(workspace);
  const ModuleWs w = carve_module_ws(wc, axvs_cc_layer_workspace_bytes(B, Q, Tc), (size_t)num_layers * R * 512 * sizeof(float), R, num_layers, 128);
  const long long mstride = (long long)B * Q * Tc * V * H * W;
  g_prof_next = 0;
  // heads of all layers (the reference's return value), or of the last one only (option "cc_last_heads_only": outputs hold one layer)
  const int hl = g_cc_last_only ? 1 : num_layers;
  const float* hq = w.q + (size_t)(num_layers - hl) * R * 256;
  auto heads = [&](hipStream_t hs) {
    float* emb = static_cast<float*>(w.heads);
    if (dtype == AXVS_BF16) {
      cc_heads_small_t<kBF>(hq, pred_logits, w.kern, packed_heads, B, Q, Tc, K1, emb, hs, hl);
      return cc_masks_t<kBF>(panoptic_features, w.kern, pred_masks, packed_heads, B, Q, Tc, V, H, W, K1, hl, R * 32, mstride, hs);
    }
    cc_heads_small_t<false>(hq, pred_logits, w.kern, packed_heads, B, Q, Tc, K1, emb, hs, hl);
    return cc_masks_t<false>(panoptic_features, w.kern, pred_masks, packed_heads, B, Q, Tc, V, H, W, K1, hl, R * 32, mstride, hs);
  };
  return run_cc_module(clip_query, packed_layers, num_layers, last_query, B, Q, Tc, rates, dtype, w, static_cast<hipStream_t>(stream), heads);
}

size_t axvs_tl_cc_module_workspace_bytes(int B, int Q, int Tc, int Cm, int num_layers) {
  Carver c(nullptr);
  Carver h(nullptr);
  carve_tl_heads_ws(h, (long long)num_layers * B * Q * Tc);
  carve_module_ws(c, axvs_cc_layer_workspace_bytes(B, Q, Tc), h.off, (long long)B * Q * Tc, num_layers, Cm);
  return c.off;
}

int axvs_tl_cc_module_fwd(const float* clip_query, const float* mask_feature, float* cls_logits, float* mask_logits, float* last_query,
                          const void* const* packed_layers, const void* packed_heads, int num_layers, int B, int Q, int Tc, int frames_per_clip,
                          int h, int w_, int K1, int Cm, const int* rates, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
  if (int rcd = check_dtype(dtype)) return rcd;
  if (!clip_query || !mask_feature || !cls_logits || !mask_logits || !last_query || !packed_layers || !packed_heads || !rates || !workspace)
    return fail(AXVS_ERR_ARG, "null pointer");
  if (B <= 0 || Q <= 0 || Tc <= 0 || frames_per_clip <= 0 || h <= 0 || w_ <= 0 || K1 <= 0 || num_layers <= 0) return fail(AXVS_ERR_ARG, "empty shape");
  if (Cm != 128 && Cm != 256) return fail(AXVS_ERR_ARG, "mask feature channels must be 128 or 256 (got %d)", Cm);
  if (Tc > 1024) return fail(AXVS_ERR_ARG, "more than 1024 clips are not supported by the class head");
  if (int rcd = check_dtype(dtype)) return rcd;
  if (workspace_bytes < axvs_tl_cc_module_workspace_bytes(B, Q, Tc, Cm, num_layers)) return fail(AXVS_ERR_WORKSPACE, "workspace too small");
  const long long R = (long long)B * Q * Tc;
  Carver hsz(nullptr);
  carve_tl_heads_ws(hsz, (long long)num_layers * R);
  Carver wc(workspace);
  const ModuleWs w = carve_module_ws(wc, axvs_cc_layer_workspace_bytes(B, Q, Tc), hsz.off, R, num_layers, Cm);
  const long long mstride = (long long)B * Tc * frames_per_clip * Q * h * w_;
  g_prof_next = 0;
  auto heads = [&](hipStream_t hs) {
    Carver hc(w.heads);
    const TLHeadsWs hw = carve_tl_heads_ws(hc, (long long)num_layers * R);
    if (dtype == AXVS_BF16) {
      tl_heads_small_t<kBF>(w.q, cls_logits, w.kern, packed_heads, B, Q, Tc, K1, Cm, hw, hs, num_layers);
      return tl_masks_t<kBF>(mask_feature, w.kern, mask_logits, B, Q, Tc, frames_per_clip, h, w_, Cm, num_layers, R * 32, mstride, hs);
    }
    tl_heads_small_t<false>(w.q, cls_logits, w.kern, packed_heads, B, Q, Tc, K1, Cm, hw, hs, num_layers);
    return tl_masks_t<false>(mask_feature, w.kern, mask_logits, B, Q, Tc, frames_per_clip, h, w_, Cm, num_layers, R * 32, mstride, hs);
  };
  return run_cc_module(clip_query, packed_layers, num_layers, last_query, B, Q, Tc, rates, dtype, w, static_cast<hipStream_t>(stream), heads);
}

size_t axvs_tl_heads_packed_bytes(int K1, int Cm) {
  Carver c(nullptr);
  carve_tl_heads(c, K1, Cm);
  return c.off;
}

int axvs_tl_heads_pack(const AxvsTLHeadParams* p, void* packed, int K1, int Cm, int dtype, void* stream) {
  if (int rcd = check_dtype(dtype)) return rcd;
  if (!p || !packed) return fail(AXVS_ERR_ARG, "null pointer");
  if (K1 <= 0 || (Cm != 128 && Cm != 256)) return fail(AXVS_ERR_ARG, "mask feature channels must be 128 or 256 (got %d)", Cm);
  if (int rcd = check_dtype(dtype)) return rcd;
  Carver c(packed);
  TLHeadsPacked h = carve_tl_heads(c, K1, Cm);
  hipStream_t st = static_cast<hipStream_t>(stream);
  PackDim n256{256, 256, 0, 0, 0}, ncm{Cm, Cm, 0, 0, 0};
  if (dtype == AXVS_BF16) {
    pack_w<kBF>(p->mask_embed_w[0], h.w0, n256, n256, st);
    pack_w<kBF>(p->mask_embed_w[1], h.w1, n256, n256, st);
    pack_w<kBF>(p->mask_embed_w[2], h.w2, ncm, n256, st);
  } else {
    pack_w<false>(p->mask_embed_w[0], h.w0, n256, n256, st);
    pack_w<false>(p->mask_embed_w[1], h.w1, n256, n256, st);
    pack_w<false>(p->mask_embed_w[2], h.w2, ncm, n256, st);
  }
  copy_f32(p->mask_embed_b[0], h.b0, 256, st);
  copy_f32(p->mask_embed_b[1], h.b1, 256, st);
  copy_f32(p->mask_embed_b[2], h.b2, Cm, st);
  copy_f32(p->post_norm_w, h.pn_w, 256, st);
  copy_f32(p->post_norm_b, h.pn_b, 256, st);
  copy_f32(p->activation_proj_w, h.wa, 256, st);
  copy_f32(p->activation_proj_b, h.ba, 1, st);
  hipLaunchKernelGGL(transpose_k1x256_kernel, dim3((unsigned)((K1 * 256 + 255) / 256)), dim3(256), 0, st, p->cls_embed_w, h.wc, K1);
  copy_f32(p->cls_embed_b, h.bc, K1, st);
  return last_launch_status();
}

size_t axvs_tl_heads_workspace_bytes(int B, int Q, int Tc, int Cm) {
  Carver c(nullptr);
  const long long R = (long long)B * Q * Tc;
  c.take<float>((size_t)R * 256);
  c.take<u16>((size_t)R * 256);
  c.take<u16>((size_t)R * 256);
  c.take<u16>((size_t)R * 256);
  c.take<u16>((size_t)R * Cm);
  return c.off;
}

int axvs_tl_heads_fwd(const float* clip_query, const float* mask_feature, float* cls_logits, float* mask_logits,
                      const void* packed, int B, int Q, int Tc, int frames_per_clip, int h, int w, int K1, int Cm, int dtype,
                      void* workspace, size_t workspace_bytes, void* stream) {
  if (int rcd = check_dtype(dtype)) return rcd;
  if (!clip_query || !mask_feature || !cls_logits || !mask_logits || !packed || !workspace) return fail(AXVS_ERR_ARG, "null pointer");
  if (B <= 0 || Q <= 0 || Tc <= 0 || frames_per_clip <= 0 || h <= 0 || w <= 0 || K1 <= 0) return fail(AXVS_ERR_ARG, "empty shape");
  if (Cm != 128 && Cm != 256) return fail(AXVS_ERR_ARG, "mask feature channels must be 128 or 256 (got %d)", Cm);
  if (Tc > 1024) return fail(AXVS_ERR_ARG, "more than 1024 clips are not supported by the class head");
  if (workspace_bytes < axvs_tl_heads_workspace_bytes(B, Q, Tc, Cm)) return fail(AXVS_ERR_WORKSPACE, "workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == AXVS_BF16) return tl_heads_fwd_t<kBF>(clip_query, mask_feature, cls_logits, mask_logits, packed, B, Q, Tc, frames_per_clip, h, w, K1, Cm, workspace, st);
  if (dtype == AXVS_F16) return tl_heads_fwd_t<false>(clip_query, mask_feature, cls_logits, mask_logits, packed, B, Q, Tc, frames_per_clip, h, w, K1, Cm, workspace, st);
  return fail(AXVS_ERR_ARG, "unknown dtype %d", dtype);
}

size_t axvs_msda_packed_bytes(int C, int heads, int L, int P) {
  Carver c(nullptr);
  carve_msda(c, C, heads, L, P);
  return c.off;
}

int axvs_msda_pack(const AxvsMsdaParams* p, void* packed, int C, int heads, int L, int P, int dtype, void* stream) {
  if (int rcd = check_dtype(dtype)) return rcd;
  if (!p || !packed) return fail(AXVS_ERR_ARG, "null pointer");
  if (int rc = check_cfg(C, heads)) return rc;
  if (L <= 0 || L > kMsdaMaxLevels || P <= 0 || L * P > 64) return fail(AXVS_ERR_ARG, "unsupported n_levels=%d / n_points=%d", L, P);
  if (int rcd = check_dtype(dtype)) return rcd;
  Carver c(packed);
  MsdaPacked m = carve_msda(c, C, heads, L, P);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int d = C / heads, Cp = heads * 32, mlp = heads * L * P;
  PackDim plainC{C, C, 0, 0, 0}, headC{C, Cp, heads, d, 0}, off{2 * mlp, 2 * mlp, 0, 0, 0}, lg{mlp, mlp, 0, 0, 0};
  if (dtype == AXVS_BF16) {
    pack_w3<kBF>(p->value_proj_w, m.wv, headC, plainC, st);
    pack_w3<kBF>(p->sampling_offsets_w, m.wq, off, plainC, st, 0, 3 * mlp);
    pack_w3<kBF>(p->attention_weights_w, m.wq, lg, plainC, st, 2 * mlp, 3 * mlp);
    pack_w3<kBF>(p->output_proj_w, m.wo, plainC, headC, st);
  } else {
    pack_w3<false>(p->value_proj_w, m.wv, headC, plainC, st);
    pack_w3<false>(p->sampling_offsets_w, m.wq, off, plainC, st, 0, 3 * mlp);
    pack_w3<false>(p->attention_weights_w, m.wq, lg, plainC, st, 2 * mlp, 3 * mlp);
    pack_w3<false>(p->output_proj_w, m.wo, plainC, headC, st);
  }
  pack_b(p->value_proj_b, m.bv, headC, st);
  copy_f32(p->sampling_offsets_b, m.bq, 2 * mlp, st);
  copy_f32(p->attention_weights_b, m.bq + 2 * mlp, mlp, st);
  copy_f32(p->output_proj_b, m.bo, C, st);
  copy_f32(p->sampling_offsets_w, m.wq32, (size_t)2 * mlp * C, st);
  copy_f32(p->attention_weights_w, m.wq32 + (size_t)2 * mlp * C, (size_t)mlp * C, st);
  copy_f32(p->value_proj_w, m.wv32, (size_t)C * C, st);
  copy_f32(p->output_proj_w, m.wo32, (size_t)C * C, st);
  return last_launch_status();
}

size_t axvs_msda_workspace_bytes(int N, int Lq, int S, int C, int heads, int L, int P) {
  Carver c(nullptr);
  const size_t Cp = (size_t)heads * 32;
  c.take<u16>((size_t)N * S * Cp);
  c.take<float>((size_t)N * Lq * 3 * heads * L * P);
  c.take<u16>(2 * (size_t)N * Lq * Cp);
  (void)C;
  return c.off;
}

int axvs_msda_fwd(const float* query, const float* reference_points, int ref_dim, const float* input_flatten,
                  const unsigned char* padding_mask, const int* spatial_shapes, float* out, const void* packed, int N, int Lq, int S,
                  int C, int heads, int L, int P, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
  if (int rcd = check_dtype(dtype)) return rcd;
  if (!query || !reference_points || !input_flatten || !spatial_shapes || !out || !packed || !workspace) return fail(AXVS_ERR_ARG, "null pointer");
  if (N <= 0 || Lq <= 0 || S <= 0 || P <= 0) return fail(AXVS_ERR_ARG, "empty shape");
  if (int rc = check_cfg(C, heads)) return rc;
  if (ref_dim != 2 && ref_dim != 4) return fail(AXVS_ERR_ARG, "Last dim of reference_points must be 2 or 4, but get %d instead.", ref_dim);
  if (L * P > 64) return fail(AXVS_ERR_ARG, "n_levels * n_points > 64 is not supported");
  if ((long long)N * S > 2147483647LL / 64 || (long long)N * Lq > 2147483647LL / 64) return fail(AXVS_ERR_ARG, "too many tokens for 32-bit row indices");
  MsdaLevels lv;
  if (int rc = msda_levels(spatial_shapes, L, S, &lv)) return rc;
  if (workspace_bytes < axvs_msda_workspace_bytes(N, Lq, S, C, heads, L, P)) return fail(AXVS_ERR_WORKSPACE, "workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  Carver pc(const_cast<void*>(packed));
  const MsdaPacked mp = carve_msda(pc, C, heads, L, P);
  if (dtype == AXVS_BF16) return msda_fwd_t<kBF>(query, reference_points, ref_dim, input_flatten, padding_mask, lv, out, mp, N, Lq, S, C, heads, P, workspace, st);
  if (dtype == AXVS_F16) return msda_fwd_t<false>(query, reference_points, ref_dim, input_flatten, padding_mask, lv, out, mp, N, Lq, S, C, heads, P, workspace, st);
  return fail(AXVS_ERR_ARG, "unknown dtype %d", dtype);
}

// ---- the two halves of the module for callers that work on the sampled rows before output_proj (Tube-Link plugin) ----
int axvs_msda_sample_fwd(const float* query, const float* query_pos, const float* reference_points, int ref_dim, const float* value,
                         const unsigned char* padding_mask, const int* spatial_shapes, float* sampled, const void* packed, int N, int Lq,
                         int S, int C, int heads, int L, int P, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
  if (int rcd = check_dtype(dtype)) return rcd;
  if (!query || !reference_points || !value || !spatial_shapes || !sampled || !packed || !workspace) return fail(AXVS_ERR_ARG, "null pointer");
  if (N <= 0 || Lq <= 0 || S <= 0 || P <= 0) return fail(AXVS_ERR_ARG, "empty shape");
  if (int rc = check_cfg(C, heads)) return rc;
  if ((C / heads) % 8) return fail(AXVS_ERR_ARG, "head_dim=%d must be a multiple of 8", C / heads);
  if (ref_dim != 2 && ref_dim != 4) return fail(AXVS_ERR_ARG, "Last dim of reference_points must be 2 or 4, but get %d instead.", ref_dim);
  if (L * P > 64) return fail(AXVS_ERR_ARG, "n_levels * n_points > 64 is not supported");
  if ((long long)N * S > 2147483647LL / 64 || (long long)N * Lq > 2147483647LL / 64) return fail(AXVS_ERR_ARG, "too many tokens for 32-bit row indices");
  MsdaLevels lv;
  if (int rc = msda_levels(spatial_shapes, L, S, &lv)) return rc;
  if (workspace_bytes < axvs_msda_workspace_bytes(N, Lq, S, C, heads, L, P)) return fail(AXVS_ERR_WORKSPACE, "workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  Carver pc(const_cast<void*>(packed));
  const MsdaPacked mp = carve_msda(pc, C, heads, L, P);
  if (dtype == AXVS_BF16) return msda_fwd_t<kBF>(query, reference_points, ref_dim, value, padding_mask, lv, sampled, mp, N, Lq, S, C, heads, P, workspace, st, query_pos, nullptr, 1);
  if (dtype == AXVS_F16) return msda_fwd_t<false>(query, reference_points, ref_dim, value, padding_mask, lv, sampled, mp, N, Lq, S, C, heads, P, workspace, st, query_pos, nullptr, 1);
  return fail(AXVS_ERR_ARG, "unknown dtype %d", dtype);
}

int axvs_msda_output_proj_fwd(const float* x, const float* identity, float* out, const void* packed, long long rows, int C, int heads,
                              int L, int P, int dtype, void* stream) {
  if (int rcd = check_dtype(dtype)) return rcd;
  if (!x || !out || !packed) return fail(AXVS_ERR_ARG, "null pointer");
  if (rows <= 0 || rows > 2147483647LL / 64) return fail(AXVS_ERR_ARG, "bad row count");
  if (int rc = check_cfg(C, heads)) return rc;
  if (C / heads != 32) return fail(AXVS_ERR_ARG, "axvs_msda_output_proj_fwd needs head_dim 32 (got %d)", C / heads);
  if (int rcd = check_dtype(dtype)) return rcd;
  hipStream_t st = static_cast<hipStream_t>(stream);
  Carver pc(const_cast<void*>(packed));
  const MsdaPacked mp = carve_msda(pc, C, heads, L, P);
  const EpiRowsF32 e{out, identity, mp.bo, identity_map(rows), C, 1.f};
  // split-precision operands as in the module path (the projection output has no norm behind it)
  if (use_nt128(rows, C, heads)) {
    tr::GemmEpi e128{mp.bo, 1.f, 0, tr::Drop{0u, 0u, 0u, 1.f}, 0.f};
    e128.res = identity;
    if (int rc = launch_nt128(x, nullptr, mp.wo32, out, rows, C, C, e128, st, true)) return rc;
    return last_launch_status();
  }
  if (dtype == AXVS_BF16) launch_gemm<kBF>(ALoadRowsF32Split3<kBF>{x, (int)rows, C}, mp.wo, e, (int)rows, C, 3 * C, st);
  else launch_gemm<false>(ALoadRowsF32Split3<false>{x, (int)rows, C}, mp.wo, e, (int)rows, C, 3 * C, st);
  return last_launch_status();
}

// ---- MSDeformAttnTransformerEncoderLayer (WC/msdeformattn.py:177-216): self-attention + residual, norm1, FFN, norm2 ----
size_t axvs_msda_layer_packed_bytes(int C, int heads, int L, int P, int d_ffn) {
  Carver c(nullptr);
  carve_msda(c, C, heads, L, P);
  carve_ffn(c, C, d_ffn);
  return c.off;
}

int axvs_msda_layer_pack(const AxvsMsdaLayerParams* p, void* packed, int C, int heads, int L, int P, int d_ffn, int dtype, void* stream) {
  if (int rcd = check_dtype(dtype)) return rcd;
  if (!p || !packed) return fail(AXVS_ERR_ARG, "null pointer");
  if (d_ffn <= 0 || d_ffn % 32 != 0) return fail(AXVS_ERR_ARG, "d_ffn=%d must be a positive multiple of 32", d_ffn);
  if (int rc = axvs_msda_pack(&p->self_attn, packed, C, heads, L, P, dtype, stream)) return rc;
  Carver c(packed);
  carve_msda(c, C, heads, L, P);
  LayerPacked l = carve_ffn(c, C, d_ffn);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == AXVS_BF16) pack_ffn<kBF>(p->norm1_w, p->norm1_b, p->linear1_w, p->linear1_b, p->linear2_w, p->linear2_b, p->norm2_w, p->norm2_b, l, C, d_ffn, st);
  else pack_ffn<false>(p->norm1_w, p->norm1_b, p->linear1_w, p->linear1_b, p->linear2_w, p->linear2_b, p->norm2_w, p->norm2_b, l, C, d_ffn, st);
  return last_launch_status();
}

size_t axvs_msda_layer_workspace_bytes(int N, int S, int C, int heads, int L, int P, int d_ffn) {
  Carver c(nullptr);
  c.take<char>(axvs_msda_workspace_bytes(N, S, S, C, heads, L, P));
  const size_t M = (size_t)N * S;
  c.take<float>(M * C);            // x = src + attention
  c.take<float>(M * C);            // generic FFN path scratch
  c.take<u16>(M * C);
  c.take<u16>(M * d_ffn);
  return c.off;
}

int axvs_msda_layer_fwd(const float* src, const float* pos, const float* reference_points, int ref_dim, const unsigned char* padding_mask,
                        const int* spatial_shapes, float* out, const void* packed, int N, int S, int C, int heads, int L, int P,
                        int d_ffn, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
  if (int rcd = check_dtype(dtype)) return rcd;
  if (!src || !reference_points || !spatial_shapes || !out || !packed || !workspace) return fail(AXVS_ERR_ARG, "null pointer");
  if (N <= 0 || S <= 0 || P <= 0) return fail(AXVS_ERR_ARG, "empty shape");
  if (int rc = check_cfg(C, heads)) return rc;
  if (ref_dim != 2 && ref_dim != 4) return fail(AXVS_ERR_ARG, "Last dim of reference_points must be 2 or 4, but get %d instead.", ref_dim);
  if (L * P > 64) return fail(AXVS_ERR_ARG, "n_levels * n_points > 64 is not supported");
  if (out == src) return fail(AXVS_ERR_ARG, "out must not alias src");
  if ((long long)N * S > 2147483647LL / 64) return fail(AXVS_ERR_ARG, "too many tokens for 32-bit row indices");
  MsdaLevels lv;
  if (int rc = msda_levels(spatial_shapes, L, S, &lv)) return rc;
  if (workspace_bytes < axvs_msda_layer_workspace_bytes(N, S, C, heads, L, P, d_ffn)) return fail(AXVS_ERR_WORKSPACE, "workspace too small");
  if (int rcd = check_dtype(dtype)) return rcd;
  hipStream_t st = static_cast<hipStream_t>(stream);
  Carver pc(const_cast<void*>(packed));
  const MsdaPacked mp = carve_msda(pc, C, heads, L, P);
  const LayerPacked lp = carve_ffn(pc, C, d_ffn);
  Carver wc(workspace);
  void* mws = wc.take<char>(axvs_msda_workspace_bytes(N, S, S, C, heads, L, P));
  const long long M = (long long)N * S;
  float* x = wc.take<float>((size_t)M * C);
  float* tmp = wc.take<float>((size_t)M * C);
  u16* y16 = wc.take<u16>((size_t)M * C);
  u16* h16 = wc.take<u16>((size_t)M * d_ffn);
  int rc = dtype == AXVS_BF16
               ? msda_fwd_t<kBF>(src, reference_points, ref_dim, src, padding_mask, lv, x, mp, N, S, S, C, heads, P, mws, st, pos, src)
               : msda_fwd_t<false>(src, reference_points, ref_dim, src, padding_mask, lv, x, mp, N, S, S, C, heads, P, mws, st, pos, src);
  if (rc != AXVS_OK) return rc;
  rc = dtype == AXVS_BF16 ? run_ffn<kBF>(x, out, lp, M, C, heads, d_ffn, tmp, y16, h16, st)
                          : run_ffn<false>(x, out, lp, M, C, heads, d_ffn, tmp, y16, h16, st);
  return rc != AXVS_OK ? rc : last_launch_status();
}

int axvs_msda_core_fwd(const float* value, const int* spatial_shapes, const float* sampling_loc, const float* attn_weight, float* out,
                       int N, int S, int M, int D, int Lq, int L, int P, void* stream) {
  if (!value || !spatial_shapes || !sampling_loc || !attn_weight || !out) return fail(AXVS_ERR_ARG, "null pointer");
  if (N <= 0 || S <= 0 || M <= 0 || D <= 0 || Lq <= 0 || P <= 0) return fail(AXVS_ERR_ARG, "empty shape");
  MsdaLevels lv;
  if (int rc = msda_levels(spatial_shapes, L, S, &lv)) return rc;
  const long long total = (long long)N * Lq * M * D;
  hipLaunchKernelGGL(msda_core_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), value, lv,
                     sampling_loc, attn_weight, out, N, S, M, D, Lq, P);
  return last_launch_status();
}

int axvs_msda_core_bwd(const float* value, const int* spatial_shapes, const float* sampling_loc, const float* attn_weight, const float* grad_output,
                       float* grad_value, float* grad_sampling_loc, float* grad_attn_weight, int N, int S, int M, int D, int Lq, int L, int P,
                       void* stream) {
  if (!value || !spatial_shapes || !sampling_loc || !attn_weight || !grad_output || !grad_value || !grad_sampling_loc || !grad_attn_weight)
    return fail(AXVS_ERR_ARG, "null pointer");
  if (N <= 0 || S <= 0 || M <= 0 || D <= 0 || Lq <= 0 || P <= 0) return fail(AXVS_ERR_ARG, "empty shape");
  MsdaLevels lv;
  if (int rc = msda_levels(spatial_shapes, L, S, &lv)) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long long total = (long long)N * Lq * M * D, samples = (long long)N * Lq * M * L * P;
  if (hipMemsetAsync(grad_value, 0, (size_t)N * S * M * D * sizeof(float), st) != hipSuccess) return fail(AXVS_ERR_LAUNCH, "hipMemsetAsync failed");
  const bool shfl = D <= 64 && (D & (D - 1)) == 0;        // a (n, q, m) group = D aligned lanes of one wave
  const dim3 grid((unsigned)((total + 255) / 256));
  if (shfl) {
    hipLaunchKernelGGL(msda_core_bwd_kernel<true>, grid, dim3(256), 0, st, value, lv, sampling_loc, attn_weight, grad_output, grad_value,
                       grad_sampling_loc, grad_attn_weight, N, S, M, D, Lq, P);
  } else {
    if (hipMemsetAsync(grad_sampling_loc, 0, (size_t)samples * 2 * sizeof(float), st) != hipSuccess ||
        hipMemsetAsync(grad_attn_weight, 0, (size_t)samples * sizeof(float), st) != hipSuccess)
      return fail(AXVS_ERR_LAUNCH, "hipMemsetAsync failed");
    hipLaunchKernelGGL(msda_core_bwd_kernel<false>, grid, dim3(256), 0, st, value, lv, sampling_loc, attn_weight, grad_output, grad_value,
                       grad_sampling_loc, grad_attn_weight, N, S, M, D, Lq, P);
  }
  return last_launch_status();
}

// ---- pixel-decoder glue ----
size_t axvs_conv1x1_gn_packed_bytes(int Cin, int Cout) {
  Carver c(nullptr);
  c.take<u16>(3 * (size_t)Cin * ((Cout + 15) & ~15));
  c.take<float>(Cout); c.take<float>(Cout); c.take<float>(Cout);
  c.take<float>((size_t)Cout * Cin);      // the fp32 weight as it is: the 128 x 128 split-precision GEMM splits its operands itself (token rows in, many rows)
  return c.off;
}

int axvs_conv1x1_gn_pack(const AxvsConvGnParams* p, void* packed, int Cin, int Cout, int dtype, void* stream) {
  if (int rcd = check_dtype(dtype)) return rcd;
  if (!p || !packed) return fail(AXVS_ERR_ARG, "null pointer");
  if (Cin <= 0 || Cin % 32 || Cout <= 0 || Cout % 4) return fail(AXVS_ERR_ARG, "Cin=%d must be a multiple of 32, Cout=%d of 4", Cin, Cout);
  if (int rcd = check_dtype(dtype)) return rcd;
  Carver c(packed);
  u16* w = c.take<u16>(3 * (size_t)Cin * ((Cout + 15) & ~15));
  float* b = c.take<float>(Cout); float* g = c.take<float>(Cout); float* be = c.take<float>(Cout);
  hipStream_t st = static_cast<hipStream_t>(stream);
  PackDim nd{Cout, Cout, 0, 0, 0}, kd{Cin, Cin, 0, 0, 0};
  if (dtype == AXVS_BF16) pack_w3<kBF>(p->conv_w, w, nd, kd, st);
  else pack_w3<false>(p->conv_w, w, nd, kd, st);
  copy_f32(p->conv_b, b, Cout, st);
  copy_f32(p->gn_w, g, Cout, st);
  copy_f32(p->gn_b, be, Cout, st);
  float* wf = c.take<float>((size_t)Cout * Cin);
  if (hipMemcpyAsync(wf, p->conv_w, (size_t)Cout * Cin * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) return fail(AXVS_ERR_LAUNCH, "copy failed");
  return last_launch_status();
}

size_t axvs_conv1x1_gn_workspace_bytes(int N, int HW, int Cout, int groups) {
  Carver c(nullptr);
  c.take<float>((size_t)N * HW * Cout);
  c.take<float>((size_t)N * HW * Cout);      // token-row copy of an NCHW input (callers size with max(Cin, Cout); without the room the NCHW loader runs)
  c.take<float>((size_t)N * groups * 2);
  c.take<float>((size_t)N * ((HW + 63) / 64) * groups * 2);
  return c.off;
}

int axvs_conv1x1_gn_fwd(const float* x, int in_layout, long long in_batch_stride, long long in_ld, float* out, int out_layout,
                        long long out_batch_stride, long long out_ld, const void* packed, int N, int HW, int Cin, int Cout, int groups,
                        float eps, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
  if (int rcd = check_dtype(dtype)) return rcd;
  if (!x || !out || !packed || !workspace) return fail(AXVS_ERR_ARG, "null pointer");
  if (N <= 0 || HW <= 0) return fail(AXVS_ERR_ARG, "empty shape");
  if (Cin % 32 || Cout % 4 || groups <= 0 || Cout % groups || groups > 256) return fail(AXVS_ERR_ARG, "unsupported channels/groups %d/%d/%d", Cin, Cout, groups);
  if ((in_layout != 0 && in_layout != 1) || (out_layout != 0 && out_layout != 1)) return fail(AXVS_ERR_ARG, "layout must be 0 (NCHW) or 1 (token rows)");
  if ((long long)N * HW > 2147483647LL / 64) return fail(AXVS_ERR_ARG, "too many tokens for 32-bit row indices");
  if (workspace_bytes < axvs_conv1x1_gn_workspace_bytes(N, HW, Cout, groups)) return fail(AXVS_ERR_WORKSPACE, "workspace too small");
  if (int rcd = check_dtype(dtype)) return rcd;
  hipStream_t st = static_cast<hipStream_t>(stream);
  Carver pc(const_cast<void*>(packed));
  const u16* w = pc.take<u16>(3 * (size_t)Cin * ((Cout + 15) & ~15));
  const float* b = pc.take<float>(Cout); const float* g = pc.take<float>(Cout); const float* be = pc.take<float>(Cout);
  const float* wf = pc.take<float>((size_t)Cout * Cin);
  Carver wc(workspace);
  const long long M = (long long)N * HW;
  float* y = wc.take<float>((size_t)M * Cout);
  float* stats = wc.take<float>((size_t)N * groups * 2);
  const int nblk = (HW + 63) / 64;
  float* partial = wc.take<float>((size_t)N * nblk * groups * 2);
  if (Cout > 8192) return fail(AXVS_ERR_ARG, "Cout=%d too large for the GroupNorm statistics kernel", Cout);
  g_prof_next = 0;
  mark(st, "begin");
  const EpiRowsF32 ey{y, nullptr, b, identity_map(M), Cout, 1.f};
  // token rows in, many of them (the output projections of the large pyramid levels: [32786 x 512 x 256] at the shipped VIPSeg setting): the 128 x 128
  // three-piece kernel of the deformable attention's projections (fp32-grade like the split3 GEMM below; 143 against 60 - 115 TFLOP/s), one launch per
  // frame when the frames are rows of a larger buffer
  // (every launch has to fill the chip on its own: >= 192 tiles of 128 x 128 -- BASELINE config 3's 64 x 64 level, 64 tiles per frame, stays on the kernels below)
  const bool one = in_batch_stride == (long long)HW * in_ld;
  // split-K factor of the NCHW path: few 128 x 128 tiles and a long reduction -> enough workgroups for one round of the chip (at most 8, at least 4 k-steps each)
  int zs = 1;
  if (in_layout == 0 && g_conv_nt128_splitk && Cin >= g_conv_nt128_splitk) {
    const long long tiles = ((M + 127) / 128) * ((Cout + 127) / 128);
    if (tiles < g_conv_nt128_nchw && tiles >= 8) {
      zs = (int)std::min<long long>(8, std::max<long long>(1, 256 / tiles));
      zs = std::min(zs, Cin / 32 / 4);
      // the partials live behind the token-row copy: as many as the caller's workspace has room for
      const size_t off_tok = axvs_conv1x1_gn_workspace_bytes(N, HW, Cout, groups) - (size_t)M * Cout * sizeof(float), tok_end = off_tok + (size_t)M * Cin * sizeof(float);
      const long long room = workspace_bytes > tok_end ? (long long)((workspace_bytes - tok_end) / ((size_t)M * Cout * sizeof(float))) : 0;
      zs = (int)std::min<long long>(zs, room);
      if (zs < 2) zs = 1;
    }
  }
  if (in_layout == 1 && g_msda_gemm && g_conv_nt128 && Cin % 4 == 0 && Cout % 4 == 0 && in_ld % 4 == 0 && in_batch_stride % 4 == 0 &&
      (((one ? M : (long long)HW) + 127) / 128) * ((Cout + 127) / 128) >= g_conv_nt128 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
    const tr::GemmEpi e{b, 1.f, 0, tr::Drop{0, 0, 0, 1.f}, 0.f};
    for (int n = 0; n < (one ? 1 : N); ++n)
      if (int rc = launch_nt128(x + (size_t)n * in_batch_stride, nullptr, wf, y + (size_t)n * HW * Cout, one ? M : HW, Cout, Cin, e, st, g_conv_nt128_exact != 0, in_ld)) return rc;
  } else if (in_layout == 0 && g_msda_gemm && g_conv_nt128 && Cin % 4 == 0 && Cout % 4 == 0 &&
             (((M + 127) / 128) * ((Cout + 127) / 128) >= g_conv_nt128_nchw || zs > 1) && (reinterpret_cast<uintptr_t>(y) & 15) == 0 &&

             workspace_bytes >= axvs_conv1x1_gn_workspace_bytes(N, HW, Cout, groups) - (size_t)M * Cout * sizeof(float) + (size_t)M * Cin * sizeof(float)) {
    // NCHW in, many rows: transposed to token rows once (64 x 64 tiles through LDS), then the same 128 x 128 kernel in ONE launch over all frames
    float* tok = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + axvs_conv1x1_gn_workspace_bytes(N, HW, Cout, groups) - (size_t)M * Cout * sizeof(float));
    hipLaunchKernelGGL(nchw_to_tokens_kernel, dim3((unsigned)((HW + 63) / 64), (unsigned)((Cin + 63) / 64), N), dim3(256), 0, st, x, tok, Cin, HW);
    if (zs > 1) {      // few row tiles, long reduction (the coarsest level: [2150 x 256 x 2048]): split-K partials behind the token rows, added in z order + bias
      float* part = tok + (size_t)M * Cin;
      const tr::GemmEpi e{nullptr, 1.f, 0, tr::Drop{0, 0, 0, 1.f}, 0.f};
      if (int rc = launch_nt128(tok, nullptr, wf, part, M, Cout, Cin, e, st, g_conv_nt128_exact != 0, 0, zs)) return rc;
      const long long tot4 = M * Cout / 4;
      hipLaunchKernelGGL(splitk_sum_bias_kernel, dim3((unsigned)((tot4 + 255) / 256)), dim3(256), 0, st, (const float*)part, y, zs, M * Cout, b, Cout, tot4);
    } else {
      const tr::GemmEpi e{b, 1.f, 0, tr::Drop{0, 0, 0, 1.f}, 0.f};
      if (int rc = launch_nt128(tok, nullptr, wf, y, M, Cout, Cin, e, st, g_conv_nt128_exact != 0)) return rc;
    }
  } else
  if (dtype == AXVS_BF16) {
    if (in_layout == 0) launch_gemm<kBF>(ALoadNCHWSplit3<kBF>{x, (int)M, Cin, HW}, w, ey, (int)M, Cout, 3 * Cin, st);
    else launch_gemm<kBF>(ALoadTokensSplit3<kBF>{x, (int)M, Cin, HW, in_batch_stride, in_ld}, w, ey, (int)M, Cout, 3 * Cin, st);
  } else {
    if (in_layout == 0) launch_gemm<false>(ALoadNCHWSplit3<false>{x, (int)M, Cin, HW}, w, ey, (int)M, Cout, 3 * Cin, st);
    else launch_gemm<false>(ALoadTokensSplit3<false>{x, (int)M, Cin, HW, in_batch_stride, in_ld}, w, ey, (int)M, Cout, 3 * Cin, st);
  }
  mark(st, "glue.conv1x1");
  {
    const int n4 = Cout / 4, lanes = n4 < 256 ? n4 : 256, rg = 256 / lanes;
    hipLaunchKernelGGL(gn_stats_kernel, dim3((unsigned)nblk, N), dim3(256), 2 * (size_t)rg * Cout * sizeof(float), st, y, partial, HW, Cout, groups);
  }
  // (the per-group sums of the blocks' partials are formed by the apply kernel itself: no launch of their own)
  (void)stats;
  const dim3 ag((unsigned)((HW + 63) / 64), (unsigned)((Cout + 63) / 64), N);
  if (out_layout == 0) hipLaunchKernelGGL((gn_apply_kernel<true>), ag, dim3(256), 0, st, y, (const float*)partial, nblk, g, be, out, HW, Cout, groups, eps, (long long)0, (long long)Cout * HW);
  else hipLaunchKernelGGL((gn_apply_kernel<false>), ag, dim3(256), 0, st, y, (const float*)partial, nblk, g, be, out, HW, Cout, groups, eps, out_ld, out_batch_stride);
  mark(st, "glue.group_norm");
  return last_launch_status();
}

int axvs_pos2d(float* pos, const float* add, int N, int H, int W, int C, long long S, long long row0, float temperature, int normalize,
               float scale, void* stream) {
  if (!pos) return fail(AXVS_ERR_ARG, "null pointer");
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 2 || row0 < 0 || row0 + (long long)H * W > S) return fail(AXVS_ERR_ARG, "bad shape");
  const long long total = (long long)H * W * C;
  hipLaunchKernelGGL(pos2d_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), pos, add, N, H, W,
                     C, S, row0, temperature, normalize, scale);
  return last_launch_status();
}

// ---- clip-to-clip query alignment ----
static int launch_lsap(const float* cost, long long* col4row, int batch, int n, hipStream_t st, int chain = 1) {
  const size_t bytes = (size_t)n * n * sizeof(float);
  const bool lds = bytes <= 128 * 1024;
#define AXVS_LSAP(CPL)                                                                                                        \
  do {                                                                                                                        \
    if (lds) {                                                                                                                \
      if (int rc = ensure_max_lds(reinterpret_cast<const void*>(&lsap_kernel<CPL, true>))) return rc;             \
      hipLaunchKernelGGL((lsap_kernel<CPL, true>), dim3(batch), dim3(64), bytes, st, cost, col4row, n, chain);                \
    } else {                                                                                                                  \
      hipLaunchKernelGGL((lsap_kernel<CPL, false>), dim3(batch), dim3(64), 0, st, cost, col4row, n, chain);                   \
    }                                                                                                                         \
  } while (0)
  if (n <= 64) AXVS_LSAP(1);
  else if (n <= 128) AXVS_LSAP(2);
  else if (n <= 256) AXVS_LSAP(4);
  else AXVS_LSAP(8);
#undef AXVS_LSAP
  return last_launch_status();
}

int axvs_linear_sum_assignment(const float* cost, long long* col4row, int batch, int n, void* stream) {
  if (!cost || !col4row) return fail(AXVS_ERR_ARG, "null pointer");
  if (batch <= 0 || n <= 0 || n > kLsapMax) return fail(AXVS_ERR_ARG, "n=%d must be in 1..%d", n, kLsapMax);
  return launch_lsap(cost, col4row, batch, n, static_cast<hipStream_t>(stream));
}

static int cost_tile_lds(int C) {     // two 16-row panels of C + 1 floats
  const size_t bytes = (size_t)2 * 16 * (C + 1) * sizeof(float);
  if (bytes > 160 * 1024) return fail(AXVS_ERR_ARG, "embedding width C=%d too large for the cost kernel's LDS panels", C);
  return bytes > 64 * 1024 ? ensure_max_lds(reinterpret_cast<const void*>(&cost_tile_kernel)) : AXVS_OK;
}

size_t axvs_match_embds_workspace_bytes(int Q, int C) { return ((size_t)Q * Q + 2 * (size_t)Q * C) * sizeof(float); }

int axvs_match_embds(const float* tgt_embds, const float* cur_embds, long long* indices, int Q, int C, void* workspace,
                     size_t workspace_bytes, void* stream) {
  if (!tgt_embds || !cur_embds || !indices || !workspace) return fail(AXVS_ERR_ARG, "null pointer");
  if (Q <= 0 || Q > kLsapMax || C <= 0) return fail(AXVS_ERR_ARG, "Q=%d must be in 1..%d", Q, kLsapMax);
  if (workspace_bytes < axvs_match_embds_workspace_bytes(Q, C)) return fail(AXVS_ERR_WORKSPACE, "workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* cost = static_cast<float*>(workspace);
  float* nrm = cost + (size_t)Q * Q;
  if (int rc = cost_tile_lds(C)) return rc;
  hipLaunchKernelGGL(normalize_rows_kernel, dim3((unsigned)((2 * Q + 3) / 4)), dim3(256), 0, st, tgt_embds, cur_embds, nrm, Q, C);
  hipLaunchKernelGGL(cost_tile_kernel, dim3((Q + 15) / 16, (Q + 15) / 16, 1), dim3(256), (size_t)2 * 16 * (C + 1) * sizeof(float), st, (const float*)nrm,
                     cost, 0, Q, C);
  return launch_lsap(cost, indices, 1, Q, st);
}

// the whole clip-alignment loop of a batch of videos (maxtron_cc_model.py:280-301) in three launches
size_t axvs_match_clips_workspace_bytes(int V, int Tc, int Q, int C) {
  return ((size_t)V * Tc * Q * C + (size_t)V * (Tc > 1 ? Tc - 1 : 1) * Q * Q) * sizeof(float);
}

int axvs_match_clips(const float* mask_embeddings, long long* indices, int V, int Tc, int Q, int C, void* workspace,
                     size_t workspace_bytes, void* stream) {
  if (!mask_embeddings || !indices || !workspace) return fail(AXVS_ERR_ARG, "null pointer");
  if (V <= 0 || Tc <= 1 || C <= 0 || Q <= 0 || Q > kLsapMax) return fail(AXVS_ERR_ARG, "need V >= 1, Tc >= 2, Q in 1..%d", kLsapMax);
  if (workspace_bytes < axvs_match_clips_workspace_bytes(V, Tc, Q, C)) return fail(AXVS_ERR_WORKSPACE, "workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* nrm = static_cast<float*>(workspace);
  float* cost = nrm + (size_t)V * Tc * Q * C;
  const long long R = (long long)V * Tc * Q;
  if (int rc = cost_tile_lds(C)) return rc;
  hipLaunchKernelGGL(normalize_rows1_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, st, mask_embeddings, nrm, R, C);
  hipLaunchKernelGGL(cost_tile_kernel, dim3((Q + 15) / 16, (Q + 15) / 16, (unsigned)(V * (Tc - 1))), dim3(256), (size_t)2 * 16 * (C + 1) * sizeof(float), st,
                     (const float*)nrm, cost, Tc, Q, C);
  return launch_lsap(cost, indices, V, Q, st, Tc - 1);
}

int axvs_add_channel_vector(float* x, const float* v, size_t n, int C, void* stream) {
  if (!x || !v || C <= 0) return fail(AXVS_ERR_ARG, "bad argument");
  hipLaunchKernelGGL(add_channel_vector_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x, v, n, C);
  return last_launch_status();
}

int axvs_pos3d(float* pos, int B, int T, int H, int W, int C, float temperature, int normalize, float scale, void* stream) {
  if (!pos) return fail(AXVS_ERR_ARG, "null pointer");
  if (B <= 0 || T <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 2) return fail(AXVS_ERR_ARG, "bad shape");
  long long total = (long long)T * H * W * C;
  hipLaunchKernelGGL(pos3d_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), pos,
                     B, T, H, W, C, temperature, normalize, scale);
  return last_launch_status();
}

int axvs_pos3d_masked(float* pos, const unsigned char* mask, int B, int T, int H, int W, int C, float temperature, int normalize,
                      float scale, void* stream) {
  if (!pos || !mask) return fail(AXVS_ERR_ARG, "null pointer");
  if (B <= 0 || T <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 2) return fail(AXVS_ERR_ARG, "bad shape");
  long long total = (long long)B * T * H * W * C;
  hipLaunchKernelGGL(pos3d_masked_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), pos, mask,
                     B, T, H, W, C, temperature, normalize, scale);
  return last_launch_status();
}

int axvs_scaled_residual(const float* a, const float* b, const float* gamma, float* out, size_t n, int C, void* stream) {
  if (!a || !b || !gamma || !out || C <= 0) return fail(AXVS_ERR_ARG, "bad argument");
  hipLaunchKernelGGL(scaled_residual_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     a, b, gamma, out, n, C);
  return last_launch_status();
}

}  // extern "C"
