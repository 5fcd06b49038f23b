/* axvs.h -- C-ABI of libaxvs.so: MI355X (gfx950) kernels for the Axial-VS / MaXTron
 * axial-trajectory-attention hot path.
 *
 * Plain pointers and sizes only; no torch types.  Every function is asynchronous on the
 * hipStream_t it is given (passed as void*), never allocates or frees device memory and
 * never synchronises: the caller owns every buffer (inputs, outputs, packed weights,
 * workspace).  Return value 0 = success, negative = error (message: axvs_last_error()).
 *
 * Reference interfaces replaced (paths relative to /root/reference):
 *   WC = MaXTron_Video-kMaX/maxtron_deeplab/modeling/within_clip_tracking_module
 *   CC = MaXTron_Video-kMaX/maxtron_deeplab/modeling/cross_clip_tracking_module
 *   TL = MaXTron_Tube-Link
 *
 * Row-major fp32 tensors use nn.Linear's [out, in] weight layout.  "dtype" selects the
 * 16-bit MFMA operand type (accumulation, softmax, LayerNorm and the residual stream are
 * always fp32): AXVS_F16 meets the 1e-3 parity bar; AXVS_BF16 trades accuracy (about 3e-3)
 * for fp32-like range.
 */
#ifndef AXVS_H
#define AXVS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AXVS_F16 0
#define AXVS_BF16 1

#define AXVS_OK 0
#define AXVS_ERR_ARG (-1)       /* bad shape / null pointer / unsupported configuration */
#define AXVS_ERR_WORKSPACE (-2) /* workspace too small */
#define AXVS_ERR_LAUNCH (-3)    /* HIP launch failure */
#define AXVS_ERR_STATE (-4)     /* an EARLIER call on this thread's status word reported AXVS_STATUS_SYNC_TIMEOUT (see below) */

/* fp32 parameters of one TrajectoryAttention (WC/temporal_attention.py:21-33,
 * TL/mmdet/models/plugins/msdeformattn_pixel_decoder.py:653-665).  Device pointers.
 * CC flavour (CC/maxtron_cross_clip_tracking_module.py:79-89): the fused qkv Linear is
 * passed as three row slices of qkv.weight / qkv.bias. */
typedef struct AxvsTrajParams {
  const float *q_w, *q_b;         /* [C,C], [C]   */
  const float *k_w, *k_b;         /* [C,C], [C]   */
  const float *v_w, *v_b;         /* [C,C], [C]   */
  const float *proj_q_w, *proj_q_b;   /* [C,C], [C]   */
  const float *proj_kv_w, *proj_kv_b; /* [2C,C], [2C] */
  const float *proj_w, *proj_b;   /* [C,C], [C]   */
} AxvsTrajParams;

/* fp32 parameters of one TemporalAxialTrajectoryAttentionLayer
 * (WC/temporal_attention.py:158-178; TL/...pixel_decoder.py:730-756). */
typedef struct AxvsAxialLayerParams {
  AxvsTrajParams height_attn, width_attn;
  const float *norm1_w, *norm1_b;       /* [C] */
  const float *linear1_w, *linear1_b;   /* [F,C], [F] */
  const float *linear2_w, *linear2_b;   /* [C,F], [C] */
  const float *norm2_w, *norm2_b;       /* [C] */
} AxvsAxialLayerParams;

int axvs_version(void);
/* 1 if the library was built with the bf16 operand tier (-DAXVS_WITH_BF16; not part of the default build since round 6: the tier does not hold the
 * 1e-3 parity bar -- every entry point refuses dtype = AXVS_BF16 with AXVS_ERR_ARG otherwise), else 0 */
int axvs_has_bf16(void);
const char* axvs_last_error(void);

/* Asynchronous condition bits.  The entry points never synchronise, so conditions only a kernel can see are OR-ed into a word
 * the caller registers (per calling thread; NULL disables) and reads back when it likes.  The word may live in device memory
 * or in PINNED HOST memory (hipHostMalloc: device-visible at the same address; the kernels touch it only when a condition
 * fires, so it costs nothing on the normal path -- profiles/r5_pinned_status_probe.txt).  With a host-readable word the
 * library FAILS LOUDLY: every later call of an axial-layer entry point on this thread returns AXVS_ERR_STATE while bit 2 is set
 * (axvs_check_status() reports the same without launching anything).  The Python modules always register a pinned word.
 *   bit 0  AXVS_STATUS_FP16_RANGE: a q/k/v operand of the C = 256 fused kernels (src or src + pos) exceeded the fp16 range
 *          (|x| > 65504) or was NaN -- with f16 MFMA operands the result would silently contain inf / NaN; use AXVS_BF16. */
#define AXVS_STATUS_FP16_RANGE 1
/*   bit 2  AXVS_STATUS_SYNC_TIMEOUT: a workgroup of a merged q/k/v + trajectory launch (axvs_set_sync_buffer) gave up waiting for
 *          its sibling row tiles (~1 s; option "sync_spin_limit" = number of polls): the sync words were not zero at launch, or the
 *          sibling tiles were not scheduled (the hand-off relies on the tiles of a sequence being dispatched in order onto a GPU
 *          that can hold them: true on a whole MI355X, not under CU masking).  The outputs of that call are INVALID and its sync
 *          words are left non-zero: synchronise, zero the words (hipMemset), clear the status word, then call again. */
#define AXVS_STATUS_SYNC_TIMEOUT 4
int axvs_set_status_buffer(int* device_word);
/* 0, or AXVS_ERR_STATE when the registered status word is host-readable and has AXVS_STATUS_SYNC_TIMEOUT set (reads the word as it
 * is: synchronise first to see launches still in flight).  A word in device memory cannot be inspected: returns 0. */
int axvs_check_status(void);

/* Synchronisation words of the ONE-LAUNCH-PER-PASS form of the axial layer (axvs_axial_layer_fwd[_sine3d], axvs_axial_pass_fwd;
 * C = 256, 8 heads, T <= 4, axis lengths up to 96 -- any length since round 5: frames are padded to a multiple of 16 rows in the
 * library's own q/k/v row space, the boundary tensors stay dense).  Reference: WC/temporal_attention.py:197-213 -- the
 * q/k/v Linear layers and the trajectory attention of a pass.  With a buffer registered, the trajectory kernel of a pass computes
 * q, k, v of its own 64 rows itself and the row tiles of one sequence hand K / V^T to each other INSIDE the launch, through one
 * arrival counter per sequence (B*W for the height pass, B*H for the width pass) taken from `device_words`; without one (the
 * default) every pass is a q/k/v launch followed by a trajectory launch.  Both forms give bit-identical results.
 * Contract: `device_words` are n_words 32-bit words of device memory that are ZERO when registered (hipMemset once); every launch
 * leaves them zero again.  One buffer serves one stream at a time: calls that may run concurrently (different streams) need
 * different buffers.  Registration is per calling thread, like the status word; (NULL, 0) unregisters.  A call with more
 * sequences than n_words runs the two-launch form. */
int axvs_set_sync_buffer(unsigned* device_words, size_t n_words);

/* ---- optional per-stage timing of axvs_axial_layer_fwd (used by bench.py; thread-local).
 *      events: array of `capacity` hipEvent_t created by the caller, or NULL to switch off.  While set, the layer
 *      records events[i] on its stream after stage i (events[0] at entry).  Returns the maximum stage count;
 *      after a forward, axvs_profile_stage_count() / _name(i) describe the stages that forward actually ran. */
int axvs_profile_stages(void** events, int capacity);
int axvs_profile_stage_count(void);
const char* axvs_profile_stage_name(int i);

/* ---- options: 15 keys (round 6; rounds 2 - 5 exposed 37, most of them planner thresholds and forms measured slower: those are constants / gone).
 *      Thread-local unless noted.  Unknown keys return AXVS_ERR_ARG.
 *      Set by the host modules around their calls:
 *        "ffn_gelu"            the layer's FFN activation is exact GELU instead of ReLU (activation = "gelu", WC/temporal_attention.py:9-17)
 *        "layer_out_dtype"     0 (default): axvs_axial_layer_fwd* / axvs_axial_pass_fwd(pass = 1) / axvs_traj_layer_fwd / axvs_ffn_fwd write fp32 rows, the reference's
 *                              type; 1 / 2: `out` is a [rows, C] f16 / bf16 map written by the epilogue of the kernel that ends the layer -- the map a batch-sharded
 *                              caller sends over xGMI (BASELINE config 5), without a cast pass; fused FFN tier, contiguous frames only
 *        "cc_aspp_affine"      the ASPP projection of a cross-clip layer is followed by a per-channel scale / shift packed into aspp_norm_w / aspp_norm_b
 *                              (eval-mode SyncBatchNorm, norm_fn = 'syncbn') instead of the channels-first LayerNorm of the shipped configs
 *        "cc_last_heads_only"  axvs_cc_module_fwd computes the predictor heads of the LAST layer only (pred_logits / pred_masks then hold one layer)
 *        "train_amp"           process-wide; 1 / 2: the X W^T GEMMs of the training tier multiply ONE bf16 / fp16 piece per operand (what torch.autocast gives nn.Linear)
 *        "no_merge_qkv"        two launches per axial pass (q/k/v kernel + trajectory kernel) instead of the merged launch with its in-launch hand-off: bit-identical;
 *                              the 'verify' hand-off policy's re-run
 *      Tier selection for parity tests and measurements:
 *        "generic_only", "no_attn_fusion", "no_ffn_fusion"   the shape-generic kernels / separate attention and temporal kernels / the FFN in its own kernel
 *        "merge_qkv_any"       merged launches at every grid size (default: up to ~2 rounds of the chip, or frames of 64 keys)
 *        "spatial_only"        timing: 1 = the fused trajectory kernels return after QK^T / softmax / AV, 2 = the merged kernels after their q/k/v part (outputs unwritten)
 *        "train_valu", "train_exact"   process-wide: VALU instead of fp32-MFMA attention in the training tier; 1 (default) / 0 / 2: three- / two-piece forward GEMMs / three-piece
 *                              input-gradient GEMMs as well
 *      Test hooks:
 *        "sync_spin_limit"     polls before a hand-off wait gives up and sets AXVS_STATUS_SYNC_TIMEOUT (default 2^22, about one second; 0 restores it)
 *        "plan_force"          ONE bit mask that forces the planner's size-dependent choices between forms that are bit-identical by construction (the bit-identity tests):
 *                              1 never the 16-row trajectory tiles | 2 / 4 the 128-row FFN tiles always / never | 8 no two-chunk FFN workgroups |
 *                              16 / 32 merged launch on 16-row tiles never / at any size | 64 generic tier without the reassociated temporal half; 0 = the planner's own choice */
int axvs_set_option(const char* key, int value);

/* ---- weight packing (once per load_state_dict; result is opaque, device-resident: 16-bit operands in MFMA-fragment order,
 *      16 rows x 32 k-values per KiB, so that a wave's fragment load reads consecutive addresses in lane order) ---- */
size_t axvs_traj_packed_bytes(int C, int heads);
int axvs_traj_pack(const AxvsTrajParams* p, void* packed, int C, int heads, int dtype, void* stream);
size_t axvs_axial_layer_packed_bytes(int C, int heads, int d_ffn);
int axvs_axial_layer_pack(const AxvsAxialLayerParams* p, void* packed, int C, int heads, int d_ffn, int dtype,
                          void* stream);

/* ---- TrajectoryAttention.forward(query, key, value, num_frames)
 *      WC/temporal_attention.py:35-76.  query/key/value/out: fp32 [S, T*L, C].
 *      space_attn: NULL or fp32 [(S*heads), T*L, T, L] (the reference's second return value). */
size_t axvs_traj_attn_workspace_bytes(int S, int T, int L, int C, int heads);
int axvs_traj_attn_fwd(const float* query, const float* key, const float* value, float* out, float* space_attn,
                       const void* packed, int S, int T, int L, int C, int heads, int dtype, void* workspace,
                       size_t workspace_bytes, void* stream);

/* ---- TemporalAxialTrajectoryAttentionLayer.forward(src, pos)
 *      WC/temporal_attention.py:187-220 (TL/...pixel_decoder.py:758-791).
 *      src/out: fp32 [(B*T), (H*W), C]; pos: fp32 [B,T,H,W,C]; out may not alias src.
 *      h_attn: NULL or fp32 [(B*W*heads), T*H, T, H]; w_attn: NULL or fp32 [(B*H*heads), T*W, T, W].
 *      (Tube-Link's `f + gamma * encoder(f)`, TL/...pixel_decoder.py:623-627: axvs_scaled_residual below.) */
size_t axvs_axial_layer_workspace_bytes(int B, int T, int H, int W, int C, int heads, int d_ffn);
int axvs_axial_layer_fwd(const float* src, const float* pos, float* out, const void* packed, int B, int T, int H,
                         int W, int C, int heads, int d_ffn, int dtype, void* workspace, size_t workspace_bytes,
                         float* h_attn, float* w_attn, void* stream);

/* ---- TemporalTrajectoryAttentionLayer.forward(src, pos)  (temporal_attn_type = "trajectory")
 *      WC/temporal_attention.py:103-155: ONE TrajectoryAttention over all T*H*W tokens of a clip (q = k = src + pos, v = src,
 *      frames of H*W keys each), residual, norm1, FFN, norm2.  src/out fp32 [(B*T), HW, C]; pos fp32 [B,T,H,W,C] (= [B, T*HW, C]).
 *      Frames of more than 256 keys take a chunked-key attention kernel with an online softmax (the reference's N x N logits
 *      would be 8.6 GB at 64 x 64).  No attention-map output (the reference returns None, None). */
typedef struct AxvsTrajLayerParams {
  AxvsTrajParams temporal_attn;
  const float *norm1_w, *norm1_b, *linear1_w, *linear1_b, *linear2_w, *linear2_b, *norm2_w, *norm2_b;
} AxvsTrajLayerParams;
size_t axvs_traj_layer_packed_bytes(int C, int heads, int d_ffn);
int axvs_traj_layer_pack(const AxvsTrajLayerParams* p, void* packed, int C, int heads, int d_ffn, int dtype, void* stream);
size_t axvs_traj_layer_workspace_bytes(int B, int T, int HW, int C, int heads, int d_ffn);
int axvs_traj_layer_fwd(const float* src, const float* pos, float* out, const void* packed, int B, int T, int HW, int C, int heads,
                        int d_ffn, int dtype, void* workspace, size_t workspace_bytes, void* stream);

/* One axial pass of the layer on a LOCAL block of the token grid -- the building block of off-axis sharding of a single clip over
 * GPUs (SURVEY 8e option ii: the height pass mixes tokens along H only, so it runs on any block of columns; the width pass, norm1,
 * FFN and norm2 run on any block of rows; between the two the blocks are exchanged with one all-to-all):
 *   pass 0: out = src + height_attn(q = k = src + pos, v = src)                     (WC/temporal_attention.py:197-204)
 *   pass 1: out = norm2(FFN(norm1(src + width_attn(q = k = src + pos, v = src))))   (:206-218)
 * src / pos / out: fp32 [B,T,H,W,C] contiguous blocks (pos: the matching block of the full embedding).  Workspace:
 * axvs_axial_layer_workspace_bytes_ex(B,T,H,W,C,heads,d_ffn,0,0). */
int axvs_axial_pass_fwd(const float* src, const float* pos, float* out, const void* packed, int pass, int B, int T, int H, int W,
                        int C, int heads, int d_ffn, int dtype, void* workspace, size_t workspace_bytes, void* stream);

/* The same layer with `pos` given as a SPECIFICATION instead of a tensor: pos = PositionEmbeddingSine3D(num_pos_feats = C/2,
 * temperature, normalize, scale)(x, fmt) in channels-last form (WC/pos_embeddings.py:86-130, mask = None) plus an optional
 * per-channel vector (the decoder's level_embed_3d[lvl], WC/msdeformattn.py:112-115).  This is what every caller of the layer
 * in the reference passes; the C = 256 / 8-head kernels then evaluate the embedding inside the q/k loaders instead of reading
 * B*T*H*W*C floats per pass from HBM (other shapes: materialised into the workspace first, same results as axvs_pos3d). */
typedef struct AxvsSinePos3D {
  float temperature;
  int normalize;
  float scale;
  const float* level_embed;   /* NULL or device fp32 [C] */
} AxvsSinePos3D;
/* Exact workspace of one call, by the kernel tier it will take (the fully fused C = 256 tier only round-trips q, k and V^T:
 * ~7x less than the upper bound axvs_axial_layer_workspace_bytes returns).  want_attn_maps: h_attn / w_attn will be non-NULL;
 * sine_pos: the call is axvs_axial_layer_fwd_sine3d.  Depends on axvs_set_option state of the calling thread. */
size_t axvs_axial_layer_workspace_bytes_ex(int B, int T, int H, int W, int C, int heads, int d_ffn, int want_attn_maps,
                                           int sine_pos);
size_t axvs_axial_layer_sine3d_workspace_bytes(int B, int T, int H, int W, int C, int heads, int d_ffn);
int axvs_axial_layer_fwd_sine3d(const float* src, const AxvsSinePos3D* pos, float* out, const void* packed, int B, int T, int H,
                                int W, int C, int heads, int d_ffn, int dtype, void* workspace, size_t workspace_bytes,
                                float* h_attn, float* w_attn, void* stream);

/* ---- the same layer on a [B*T, H*W, C] VIEW whose frames are `frame_stride_rows` rows (of C floats) apart -- one level of the pixel
 *      decoder's concatenated [B*T, sum(H_l W_l), C] token buffer (WC/msdeformattn.py:258-264 splits the levels out, runs the
 *      temporal encoder on each and concatenates them again; here the level is read and written where it lies).  `out` may be
 *      `src` (in place).  Generated positions only, no attention maps; fused tier only (axvs_axial_layer_strided_ok). */
int axvs_axial_layer_strided_ok(int C, int heads, int d_ffn);
size_t axvs_axial_layer_workspace_bytes_strided(int B, int T, int H, int W, int C, int heads, int d_ffn, long long frame_stride_rows);
int axvs_axial_layer_fwd_sine3d_strided(const float* src, const AxvsSinePos3D* pos, float* out, const void* packed, int B, int T, int H,
                                        int W, int C, int heads, int d_ffn, int dtype, long long frame_stride_rows, void* workspace,
                                        size_t workspace_bytes, void* stream);

/* ---- the layer's feed-forward tail alone: out = norm2(y + linear2(relu(linear1(y)))), y = norm1(x)
 *      WC/temporal_attention.py:181-185 + :217.  x/out fp32 [M, C]; weights from a packed AxvsAxialLayerParams. */
size_t axvs_ffn_workspace_bytes(long long M, int C, int d_ffn);
int axvs_ffn_fwd(const float* x, float* out, const void* packed_layer, long long M, int C, int heads, int d_ffn,
                 int dtype, void* workspace, size_t workspace_bytes, void* stream);

/* =====================================================================================================
 * Cross-clip tracking module (CC/maxtron_cross_clip_tracking_module.py), d_model = 256, 8 heads, norm_fn = 'ln'.
 * ===================================================================================================== */
typedef struct AxvsBN { const float *w, *b, *mean, *var; } AxvsBN;   /* eval-mode (Sync)BatchNorm, eps = 1e-3 */

/* one iteration of the layer loop, CC/...:286-297 */
typedef struct AxvsCCLayerParams {
  AxvsTrajParams attn;                    /* TrajectoryAttention (:78-130); qkv passed as three row slices      */
  const float *norm_w, *norm_b;           /* TrajectoryAttentionLayer.norm, LayerNorm eps 1e-5 (:140,:156-161)  */
  const float *aspp_w[3], *aspp_b[3];     /* ASPP._aspp_conv{0,1,2}: Conv1d(256,256,3) weights [256,256,3]      */
  const float *aspp_proj_w;               /* ASPP._proj_conv_bn_act.conv: [256,768,1], no bias                  */
  const float *aspp_norm_w, *aspp_norm_b; /* ... .norm: channels-first LayerNorm eps 1e-6                       */
  const float *conv_norm_w, *conv_norm_b; /* conv_norms[i]: LayerNorm eps 1e-5 (:263,:293-295)                  */
} AxvsCCLayerParams;

size_t axvs_cc_layer_packed_bytes(void);
int axvs_cc_layer_pack(const AxvsCCLayerParams* p, void* packed, int dtype, void* stream);
size_t axvs_cc_layer_workspace_bytes(int B, int Q, int Tc);
/* clip_query / out: fp32 [B, Q, Tc, 256] (out may not alias clip_query); rates: the three atrous rates */
int axvs_cc_layer_fwd(const float* clip_query, float* out, const void* packed, int B, int Q, int Tc, const int* rates,
                      int dtype, void* workspace, size_t workspace_bytes, void* stream);

/* embedding projections + MaXTronCCPredictor (eval branch), CC/...:45-75, :266-270, :300-309 */
typedef struct AxvsCCHeadParams {
  const float* class_proj_w; AxvsBN class_proj_bn;   /* _class_embedding_projection: conv [256,256,1] + BN + GELU */
  const float* mask_proj_w;  AxvsBN mask_proj_bn;    /* _mask_embedding_projection                                */
  const float* mask_head_w;  AxvsBN mask_head_bn;    /* _predictor._transformer_mask_head: conv [128,256,1] + BN  */
  const float *class_head_w, *class_head_b;          /* _predictor._transformer_class_head: [K1,256,1], [K1]      */
  const float *act_head_w, *act_head_b;              /* _predictor._transformer_class_activation_head: [1,256,1]  */
  AxvsBN pixel_bn;                                   /* _predictor._pixel_space_mask_batch_norm (1 channel)       */
} AxvsCCHeadParams;

size_t axvs_cc_heads_packed_bytes(int K1);
int axvs_cc_heads_pack(const AxvsCCHeadParams* p, void* packed, int K1, int dtype, void* stream);
size_t axvs_cc_heads_workspace_bytes(int B, int Q, int Tc);
/* clip_query fp32 [B,Q,Tc,256]; panoptic_features fp32 [B,128,Tc*V,H,W];
 * pred_logits fp32 [1,Q,K1] (softmax pooling runs over all B*Tc entries, like the reference's dim-0 softmax);
 * pred_masks fp32 [B,Q,Tc*V,H,W] (any V*H*W: rows of pixels may start at any 4-byte boundary). */
int axvs_cc_heads_fwd(const float* clip_query, const float* panoptic_features, float* pred_logits, float* pred_masks,
                      const void* packed, int B, int Q, int Tc, int V, int H, int W, int K1, int dtype, void* workspace,
                      size_t workspace_bytes, void* stream);

/* ---- Tube-Link flavour of the cross-clip heads.  The cross-clip layers themselves are axvs_cc_layer_*:
 *      TL/models/video/tube_link_vis/mask2former_video_cc_head.py:927-946 is the same computation as CC/...:286-297.
 *      Replaces forward_head_clips (:761-781) + pred_class (:783-797) for ONE decoder layer's output. */
typedef struct AxvsTLHeadParams {
  const float *post_norm_w, *post_norm_b;               /* transformer_decoder.post_norm: LayerNorm(256), eps 1e-5 (:767) */
  const float *activation_proj_w, *activation_proj_b;   /* Linear(256,1) (:394, :791)                                      */
  const float *cls_embed_w, *cls_embed_b;               /* Linear(256,K1) (:365, :796)                                     */
  const float* mask_embed_w[3];                         /* Linear(256,256), Linear(256,256), Linear(256,Cm) (:368-371)     */
  const float* mask_embed_b[3];
} AxvsTLHeadParams;

size_t axvs_tl_heads_packed_bytes(int K1, int Cm);
int axvs_tl_heads_pack(const AxvsTLHeadParams* p, void* packed, int K1, int Cm, int dtype, void* stream);
size_t axvs_tl_heads_workspace_bytes(int B, int Q, int Tc, int Cm);
/* clip_query fp32 [B,Q,Tc,256] (a cross-clip layer's output, axvs_cc_layer_fwd layout); mask_feature fp32 [B,Tc*fpc,Cm,h,w];
 * cls_logits fp32 [B,Q,K1]; mask_logits fp32 [B,Tc*fpc,Q,h,w].  Cm in {128,256}; any h*w. */
int axvs_tl_heads_fwd(const float* clip_query, const float* mask_feature, float* cls_logits, float* mask_logits,
                      const void* packed, int B, int Q, int Tc, int frames_per_clip, int h, int w, int K1, int Cm, int dtype,
                      void* workspace, size_t workspace_bytes, void* stream);

/* ---- The whole layer loop of the cross-clip modules in one call: CrossClipTrackingModule.forward's loop
 *      (CC/maxtron_cross_clip_tracking_module.py:283-318) and Tube-Link's (TLCC:925-950 + forward_head_clips :761-781 + pred_class
 *      :783-797).  packed_layers: HOST array of num_layers device pointers (axvs_cc_layer_pack); packed_heads: axvs_cc_heads_pack /
 *      axvs_tl_heads_pack.  Outputs for EVERY layer (the reference collects them for aux_outputs): pred_logits fp32
 *      [num_layers][1 | B][Q][K1], pred_masks fp32 [num_layers][...one layer's mask tensor...]; last_query fp32 [B,Q,Tc,256].
 *      The layer chain runs first; the predictor heads share their weights across layers, so the class logits and mask kernels
 *      of all layers are one launch per GEMM and the mask einsum reads the pixel features once for all layers.  */
size_t axvs_cc_module_workspace_bytes(int B, int Q, int Tc, int num_layers);
int axvs_cc_module_fwd(const float* clip_query, const float* panoptic_features, float* pred_logits, float* pred_masks,
                       float* last_query, const void* const* packed_layers, const void* packed_heads, int num_layers, int B,
                       int Q, int Tc, int V, int H, int W, int K1, const int* rates, int dtype, void* workspace,
                       size_t workspace_bytes, void* stream);
size_t axvs_tl_cc_module_workspace_bytes(int B, int Q, int Tc, int Cm, int num_layers);
int axvs_tl_cc_module_fwd(const float* clip_query, const float* mask_feature, float* cls_logits, float* mask_logits,
                          float* last_query, const void* const* packed_layers, const void* packed_heads, int num_layers, int B,
                          int Q, int Tc, int frames_per_clip, int h, int w, int K1, int Cm, const int* rates, int dtype,
                          void* workspace, size_t workspace_bytes, void* stream);

/* ---- Multi-scale deformable attention forward.
 *      OPS = MaXTron_Video-kMaX/maxtron_deeplab/modeling/within_clip_tracking_module/ops
 *      axvs_msda_fwd  replaces MSDeformAttn.forward (OPS/modules/ms_deform_attn.py:81-125): value_proj (+ padding mask),
 *      sampling_offsets, attention_weights + softmax, sampling locations, the gather, output_proj.
 *      axvs_msda_core_fwd replaces the native op ms_deform_attn_forward (OPS/src/ms_deform_attn.h:24-47,
 *      OPS/src/cuda/ms_deform_attn_cuda.cu:21-86) with its fp32 tensors as they are; level_start_index follows from
 *      spatial_shapes and im2col_step has no meaning here. */
typedef struct AxvsMsdaParams {        /* fp32 device pointers, nn.Linear layout [out,in] (OPS/modules/ms_deform_attn.py:59-62) */
  const float *value_proj_w, *value_proj_b;                 /* [C,C], [C]                  */
  const float *sampling_offsets_w, *sampling_offsets_b;     /* [heads*L*P*2, C], [..]      */
  const float *attention_weights_w, *attention_weights_b;   /* [heads*L*P, C], [..]        */
  const float *output_proj_w, *output_proj_b;               /* [C,C], [C]                  */
} AxvsMsdaParams;

size_t axvs_msda_packed_bytes(int C, int heads, int L, int P);
int axvs_msda_pack(const AxvsMsdaParams* p, void* packed, int C, int heads, int L, int P, int dtype, void* stream);
size_t axvs_msda_workspace_bytes(int N, int Lq, int S, int C, int heads, int L, int P);
/* query fp32 [N,Lq,C]; reference_points fp32 [N,Lq,L,ref_dim], ref_dim 2 or 4; input_flatten fp32 [N,S,C];
 * padding_mask: NULL or bytes [N,S], non-zero = padding (value rows zeroed); spatial_shapes: HOST ints [L][2] = (H_l, W_l),
 * sum H_l*W_l == S; out fp32 [N,Lq,C]. */
int axvs_msda_fwd(const float* query, const float* reference_points, int ref_dim, const float* input_flatten,
                  const unsigned char* padding_mask, const int* spatial_shapes, float* out, const void* packed, int N, int Lq,
                  int S, int C, int heads, int L, int P, int dtype, void* workspace, size_t workspace_bytes, void* stream);
/* The module in two halves, for the Tube-Link plugin MultiScaleDeformableAxialTrajectoryAttention.forward
 * (TL/mmdet/models/plugins/msdeformattn_pixel_decoder.py:566-638), which runs its temporal encoder on the sampled rows of the
 * coarsest levels BEFORE output_proj (:613-633):
 *   axvs_msda_sample_fwd       :594-611  value_proj(value) (+ key_padding_mask), sampling_offsets / attention_weights of
 *                                        (query + query_pos), softmax, bilinear gather -> sampled fp32 [N,Lq,C]
 *   axvs_msda_output_proj_fwd  :633-638  out = output_proj(x) + identity (identity may be NULL); head_dim 32 only.
 * query_pos: NULL or fp32 [N,Lq,C].  Workspace: axvs_msda_workspace_bytes. */
int axvs_msda_sample_fwd(const float* query, const float* query_pos, const float* reference_points, int ref_dim,
                         const float* value, const unsigned char* padding_mask, const int* spatial_shapes, float* sampled,
                         const void* packed, int N, int Lq, int S, int C, int heads, int L, int P, int dtype, void* workspace,
                         size_t workspace_bytes, void* stream);
int axvs_msda_output_proj_fwd(const float* x, const float* identity, float* out, const void* packed, long long rows, int C,
                              int heads, int L, int P, int dtype, void* stream);
/* MSDeformAttnTransformerEncoderLayer.forward (WC/msdeformattn.py:177-216), eval:
 *   x = norm1(src + MSDeformAttn(src + pos, reference_points, src, ...));  out = norm2(x + linear2(relu(linear1(x)))) */
typedef struct AxvsMsdaLayerParams {
  AxvsMsdaParams self_attn;
  const float *norm1_w, *norm1_b, *linear1_w, *linear1_b, *linear2_w, *linear2_b, *norm2_w, *norm2_b;
} AxvsMsdaLayerParams;
size_t axvs_msda_layer_packed_bytes(int C, int heads, int L, int P, int d_ffn);
int axvs_msda_layer_pack(const AxvsMsdaLayerParams* p, void* packed, int C, int heads, int L, int P, int d_ffn, int dtype,
                         void* stream);
size_t axvs_msda_layer_workspace_bytes(int N, int S, int C, int heads, int L, int P, int d_ffn);
/* src / out fp32 [N,S,C] (out must not alias src); pos fp32 [N,S,C] or NULL; reference_points fp32 [N,S,L,ref_dim] */
int axvs_msda_layer_fwd(const float* src, const float* pos, const float* reference_points, int ref_dim,
                        const unsigned char* padding_mask, const int* spatial_shapes, float* out, const void* packed, int N,
                        int S, int C, int heads, int L, int P, int d_ffn, int dtype, void* workspace, size_t workspace_bytes,
                        void* stream);
/* value fp32 [N,S,M,D]; sampling_loc fp32 [N,Lq,M,L,P,2]; attn_weight fp32 [N,Lq,M,L,P]; out fp32 [N,Lq,M*D] */
int axvs_msda_core_fwd(const float* value, const int* spatial_shapes, const float* sampling_loc, const float* attn_weight,
                       float* out, int N, int S, int M, int D, int Lq, int L, int P, void* stream);
/* backward of the native op: replaces ms_deform_attn_backward (OPS/src/ms_deform_attn.h:49-67, OPS/src/cuda/ms_deform_attn_cuda.cu:
 * 89-157, the ms_deformable_col2im kernels of OPS/src/cuda/ms_deform_im2col_cuda.cuh).  grad_output fp32 [N,Lq,M*D] ->
 * grad_value fp32 [N,S,M,D] (zeroed here, accumulated with atomic adds like the reference: the summation order is not fixed),
 * grad_sampling_loc fp32 [N,Lq,M,L,P,2], grad_attn_weight fp32 [N,Lq,M,L,P] (written; for D not a power of two <= 64 accumulated). */
int axvs_msda_core_bwd(const float* value, const int* spatial_shapes, const float* sampling_loc, const float* attn_weight,
                       const float* grad_output, float* grad_value, float* grad_sampling_loc, float* grad_attn_weight, int N, int S, int M,
                       int D, int Lq, int L, int P, void* stream);

/* ---- Pixel-decoder glue: nn.Sequential(Conv2d(k=1), GroupNorm) between backbone NCHW maps and token rows
 *      (WC/msdeformattn.py:349-375 input_proj / output_proj; used at :412 and :434), PositionEmbeddingSine
 *      (WC/pos_embeddings.py:12-53) in token form. */
typedef struct AxvsConvGnParams {
  const float *conv_w, *conv_b;   /* Conv2d weight [Cout,Cin,1,1], bias [Cout] */
  const float *gn_w, *gn_b;       /* GroupNorm affine [Cout]                    */
} AxvsConvGnParams;
size_t axvs_conv1x1_gn_packed_bytes(int Cin, int Cout);
int axvs_conv1x1_gn_pack(const AxvsConvGnParams* p, void* packed, int Cin, int Cout, int dtype, void* stream);
size_t axvs_conv1x1_gn_workspace_bytes(int N, int HW, int Cout, int groups);
/* layouts: 0 = NCHW fp32 [N,C,H*W] (contiguous; strides ignored), 1 = token rows: row (n,p) at base + n*batch_stride + p*ld
 * (elements), so a level slice of a concatenated [N,S,C] buffer can be read / written in place. */
int axvs_conv1x1_gn_fwd(const float* x, int in_layout, long long in_batch_stride, long long in_ld, float* out, int out_layout,
                        long long out_batch_stride, long long out_ld, const void* packed, int N, int HW, int Cin, int Cout,
                        int groups, float eps, int dtype, void* workspace, size_t workspace_bytes, void* stream);
/* The same projection in train() mode under autograd (round 6): forward on the raw fp32 parameters (three-piece bf16 GEMM: fp32 accuracy; GroupNorm statistics
 * in fp32, fixed summation order) and backward (d_gamma, d_beta, d_W, d_b, d_x).  `saved` travels from the forward to the backward call (conv output, statistics,
 * the input as token rows when it was not); `scratch` is free between calls.  Layouts / strides as above; Cin, Cout multiples of 8.
 * An input gradient is produced in NCHW or contiguous token rows (d_x may be NULL). */
typedef struct AxvsConvGnGrads { float *conv_w, *conv_b, *gn_w, *gn_b; } AxvsConvGnGrads;   /* field order of AxvsConvGnParams */
size_t axvs_conv1x1_gn_train_saved_bytes(int N, int HW, int Cin, int Cout, int groups, int in_layout, long long in_batch_stride, long long in_ld);
size_t axvs_conv1x1_gn_train_scratch_bytes(int N, int HW, int Cin, int Cout, int groups, int backward);
int axvs_conv1x1_gn_train_fwd(const float* x, int in_layout, long long in_batch_stride, long long in_ld, float* out, int out_layout,
                              long long out_batch_stride, long long out_ld, const AxvsConvGnParams* p, int N, int HW, int Cin, int Cout, int groups,
                              float eps, void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes, void* stream);
int axvs_conv1x1_gn_train_bwd(const float* d_out, int out_layout, long long out_batch_stride, long long out_ld, const float* x, int in_layout,
                              long long in_batch_stride, long long in_ld, const AxvsConvGnParams* p, const AxvsConvGnGrads* grads, float* d_x, int N, int HW,
                              int Cin, int Cout, int groups, void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes, void* stream);
/* x[i] += v[i % C] in place (level_embed_3d on a channels-last position embedding, WC/msdeformattn.py:117-118) */
int axvs_add_channel_vector(float* x, const float* v, size_t n, int C, void* stream);
/* pos[n][row0 + y*W + x][c] (+ add[c] if add != NULL) inside a fp32 [N][S][C] buffer */
int axvs_pos2d(float* pos, const float* add, int N, int H, int W, int C, long long S, long long row0, float temperature,
               int normalize, float scale, void* stream);

/* ---- Clip-to-clip query alignment, maxtron_cc_model.py:360-369 `match_from_embds` without the host round trip.
 *      axvs_linear_sum_assignment: `batch` square problems, cost fp32 [batch][n][n] -> col4row int64 [batch][n] (the column
 *      assigned to every row) = scipy.optimize.linear_sum_assignment(cost)[1]; n <= 512.
 *      axvs_match_embds: tgt/cur fp32 [Q,C] -> indices int64 [Q] such that cur[indices] aligns with tgt
 *      (cost = 1 - cosine similarity, rows = target queries). */
int axvs_linear_sum_assignment(const float* cost, long long* col4row, int batch, int n, void* stream);
size_t axvs_match_embds_workspace_bytes(int Q, int C);
int axvs_match_embds(const float* tgt_embds, const float* cur_embds, long long* indices, int Q, int C, void* workspace,
                     size_t workspace_bytes, void* stream);
/* The per-video clip loop of maxtron_cc_model.py:280-301 for a batch of videos in three launches:
 *   prev = e[v,0];  for i in 1..Tc-1:  idx = match_from_embds(prev, e[v,i]);  prev = e[v,i][idx];  indices[v,i-1,:] = idx
 * mask_embeddings fp32 [V,Tc,Q,C]; indices int64 [V,Tc-1,Q] (the permutation that aligns clip i to the aligned clip i-1). */
size_t axvs_match_clips_workspace_bytes(int V, int Tc, int Q, int C);
int axvs_match_clips(const float* mask_embeddings, long long* indices, int V, int Tc, int Q, int C, void* workspace,
                     size_t workspace_bytes, void* stream);

/* ---- PositionEmbeddingSine3D.forward(x, mask=None) in channels-last form
 *      WC/pos_embeddings.py:86-130: pos fp32 [B,T,H,W,C], C = 2*num_pos_feats. */
int axvs_pos3d(float* pos, int B, int T, int H, int W, int C, float temperature, int normalize, float scale,
               void* stream);

/* The same with a padding mask (WC/pos_embeddings.py:96-106: coordinates = not_mask.cumsum along t / h / w, normalised by the
 * count over the whole axis).  mask: uint8 [B,T,H,W], non-zero = padded position. */
int axvs_pos3d_masked(float* pos, const unsigned char* mask, int B, int T, int H, int W, int C, float temperature, int normalize,
                      float scale, void* stream);

/* ---- out[i] = a[i] + gamma[i % C] * b[i]   (Tube-Link residual, TL/...pixel_decoder.py:625-627) */
int axvs_scaled_residual(const float* a, const float* b, const float* gamma, float* out, size_t n, int C,
                         void* stream);

/* =====================================================================================================
 * Training tier of the layer: forward that keeps its activations + backward.
 * Semantics: TemporalAxialTrajectoryAttentionLayer.forward in train() mode, WC/temporal_attention.py:187-220 under autograd:
 * dropout(p_dropout) on the spatial attention maps (:32, :55 -- the layer passes `dropout` as the attention's attn_drop, :164-165),
 * dropout1(p_attn_drop) on both pass outputs (:166, :204, :213), dropout2 / dropout3(p_dropout) in the FFN (:172-174, :182-183).
 * fp32 activations in natural [B,T,H,W,C] order; the Linear layers run on split-precision bf16 MFMA GEMMs (axvs_train_gemm.h,
 * no vendor BLAS); head_dim in {8,16,32,64}; T <= 16.
 * Dropout masks are a pure function of (seed, site, element offset in the reference's tensor at that site):
 *   h = seed ^ (site * 0x9E3779B9);  h = fmix32(h ^ lo32(idx));  h = fmix32(h ^ hi32(idx));  keep iff (h >> 8) >= floor(p * 2^24)
 *   (fmix32 = MurmurHash3's finaliser); sites: 1 height attention map [(B W) heads, T H, T, H], 2 height pass output [(B W), T H, C],
 *   3 width attention map [(B H) heads, T W, T, W], 4 width pass output [(B H), T W, C], 5 FFN hidden [M, F], 6 FFN output [M, C].
 * so backward (and a recomputed forward, and a CPU oracle) regenerate them: nothing but `seed` is kept.
 * ===================================================================================================== */
typedef struct AxvsTrajGrads {
  float *q_w, *q_b, *k_w, *k_b, *v_w, *v_b, *proj_q_w, *proj_q_b, *proj_kv_w, *proj_kv_b, *proj_w, *proj_b;
} AxvsTrajGrads;
typedef struct AxvsAxialLayerGrads {   /* same field order as AxvsAxialLayerParams; every buffer is WRITTEN (not accumulated) */
  AxvsTrajGrads height_attn, width_attn;
  float *norm1_w, *norm1_b, *linear1_w, *linear1_b, *linear2_w, *linear2_b, *norm2_w, *norm2_b;
} AxvsAxialLayerGrads;
/* `saved`: the activations backward needs (kept between the two calls, or rebuilt by backward when recompute != 0: then any
 * buffer of that size will do); `scratch`: temporaries of one call (backward != 0: size for axvs_axial_layer_train_bwd). */
size_t axvs_axial_layer_train_saved_bytes(int B, int T, int H, int W, int C, int heads, int d_ffn);
size_t axvs_axial_layer_train_scratch_bytes(int B, int T, int H, int W, int C, int heads, int d_ffn, int backward);
/* src fp32 [(B T),(H W),C]; pos fp32 [B,T,H,W,C]; out like src.  params: the fp32 nn.Parameter storages themselves. */
int axvs_axial_layer_train_fwd(const float* src, const float* pos, float* out, const AxvsAxialLayerParams* params, int B, int T, int H,
                               int W, int C, int heads, int d_ffn, float p_dropout, float p_attn_drop, unsigned seed, void* saved,
                               size_t saved_bytes, void* scratch, size_t scratch_bytes, void* stream);
/* d_out: gradient of `out`.  Writes d_src, d_pos (NULL: not wanted) and every buffer of `grads`.  recompute != 0: `saved` is
 * rebuilt from (src, pos, params, seed) first -- the forward pass then only has to keep its inputs. */
int axvs_axial_layer_train_bwd(const float* d_out, const float* src, const float* pos, const AxvsAxialLayerParams* params,
                               const AxvsAxialLayerGrads* grads, float* d_src, float* d_pos, int B, int T, int H, int W, int C, int heads,
                               int d_ffn, float p_dropout, float p_attn_drop, unsigned seed, int recompute, void* saved, size_t saved_bytes,
                               void* scratch, size_t scratch_bytes, void* stream);

/* =====================================================================================================
 * Training tier of the cross-clip tracking module: CrossClipTrackingModule.forward in train() mode,
 * CC/maxtron_cross_clip_tracking_module.py:275-322 under autograd -- the module the reference trains on its own with the
 * segmenter frozen (maxtron_cc_model.py:104-108).  Per layer: TrajectoryAttentionLayer (:133-173; attention-map dropout
 * p_attn_drop, site 10 + 2 l, index as in the layer above), ASPP (:176-201; _proj_drop p_aspp_drop, site 11 + 2 l, element
 * index of the reference's [(B Q), 256, Tc] tensor) + residual + LayerNorm (:293-295); then for EVERY layer the embedding
 * projections and the predictor's training branch (:300-309, :45-57) with (Sync)BatchNorm on BATCH statistics (eps 1e-3).
 * Sizes: C = 256, 8 heads, mask channels 128, norm_fn 'ln', kernel sizes 3; Q % 8 == 0, Tc <= 16, any V*H*W (the shipped
 * training setting: 12 clips of 2 frames, 193 x 337 pixel features).
 *
 * SyncBatchNorm: `allreduce` (NULL in a single process) is called with a device buffer of partial sums (and the row count)
 * that it must SUM over the ranks in place, ordered on `stream` -- three times per forward and three times per backward call.
 * The parameter gradients written are this rank's own share (like nn.SyncBatchNorm under DDP: ranks are averaged afterwards).
 * bn_stats (forward, out): [class_proj [nl][2][256] | mask_proj [nl][2][256] | mask_head [nl][2][128] | pixel [nl][2]] = the batch
 * mean and UNBIASED variance of each layer's call, for the caller's running-statistics update (momentum 0.01, in layer order).
 * The running means are only read (as the shift of the variance sums); gradients are WRITTEN, not accumulated; no gradient is
 * produced for panoptic_features (the frozen segmenter's output).
 * ===================================================================================================== */
typedef struct AxvsCCLayerGrads {     /* field order of AxvsCCLayerParams */
  AxvsTrajGrads attn;
  float *norm_w, *norm_b;
  float *aspp_w[3], *aspp_b[3];
  float *aspp_proj_w;
  float *aspp_norm_w, *aspp_norm_b;
  float *conv_norm_w, *conv_norm_b;
} AxvsCCLayerGrads;
typedef struct AxvsBNGrads { float *w, *b; } AxvsBNGrads;
typedef struct AxvsCCHeadGrads {      /* field order of AxvsCCHeadParams */
  float* class_proj_w; AxvsBNGrads class_proj_bn;
  float* mask_proj_w;  AxvsBNGrads mask_proj_bn;
  float* mask_head_w;  AxvsBNGrads mask_head_bn;
  float *class_head_w, *class_head_b;
  float *act_head_w, *act_head_b;
  AxvsBNGrads pixel_bn;
} AxvsCCHeadGrads;
typedef int (*axvs_allreduce_fn)(void* user, float* device_buf, size_t n, void* stream);   /* 0 = success */
typedef struct AxvsCCTrainCfg {
  int B, Q, Tc, V, H, W, K1, num_layers;
  int rates[3];
  float p_attn_drop, p_aspp_drop;
  unsigned seed;
  axvs_allreduce_fn allreduce;
  void* allreduce_user;
  int chain_only;            /* != 0: buffers and checks for axvs_cc_layers_train_* only (any Q; V, H, W, K1 ignored) */
} AxvsCCTrainCfg;
size_t axvs_cc_module_train_saved_bytes(const AxvsCCTrainCfg* cfg);
size_t axvs_cc_module_train_scratch_bytes(const AxvsCCTrainCfg* cfg, int backward);
size_t axvs_cc_module_train_bn_stats_floats(const AxvsCCTrainCfg* cfg);
/* clip_query fp32 [B,Q,Tc,256]; panoptic_features fp32 [B,128,Tc*V,H,W]; pred_logits fp32 [nl,1,Q,K1]; pred_masks fp32
 * [nl,B,Q,Tc*V,H,W]; layers: array of num_layers parameter sets (the fp32 nn.Parameter storages themselves; AxvsBN.mean / .var =
 * the running buffers). */
int axvs_cc_module_train_fwd(const float* clip_query, const float* panoptic_features, float* pred_logits, float* pred_masks, float* bn_stats,
                             const AxvsCCLayerParams* layers, const AxvsCCHeadParams* heads, const AxvsCCTrainCfg* cfg, void* saved,
                             size_t saved_bytes, void* scratch, size_t scratch_bytes, void* stream);
/* d_logits / d_masks: gradients of pred_logits / pred_masks (all layers: the auxiliary outputs are part of the loss, CC:311-318).
 * `saved` as the forward call left it.  Writes d_clip_query [B,Q,Tc,256] and every buffer of layer_grads[0..nl) / head_grads. */
int axvs_cc_module_train_bwd(const float* d_logits, const float* d_masks, const float* clip_query, const float* panoptic_features,
                             const AxvsCCLayerParams* layers, const AxvsCCHeadParams* heads, const AxvsCCLayerGrads* layer_grads,
                             const AxvsCCHeadGrads* head_grads, float* d_clip_query, const AxvsCCTrainCfg* cfg, void* saved, size_t saved_bytes,
                             void* scratch, size_t scratch_bytes, void* stream);

/* The layer chain alone (TrajectoryAttentionLayer + ASPP + norms of every layer, no prediction heads): Tube-Link's
 * Mask2FormerVideoCCHeadTube trains its own heads around the same layers (TL/models/video/tube_link_vis/mask2former_video_cc_head.py:
 * 925-947).  cfg: B, Q, Tc, num_layers, rates, dropouts, seed, chain_only = 1 (no all-reduce: the chain has no BatchNorm); buffers
 * sized by axvs_cc_module_train_saved_bytes / _scratch_bytes of that cfg.  out_queries / d_queries: fp32 [nl,B,Q,Tc,256] -- the clip
 * queries after every layer / the gradient that reaches each of them from outside the chain. */
int axvs_cc_layers_train_fwd(const float* clip_query, float* out_queries, const AxvsCCLayerParams* layers, const AxvsCCTrainCfg* cfg, void* saved,
                             size_t saved_bytes, void* scratch, size_t scratch_bytes, void* stream);
int axvs_cc_layers_train_bwd(const float* d_queries, const float* clip_query, const AxvsCCLayerParams* layers, const AxvsCCLayerGrads* layer_grads,
                             float* d_clip_query, const AxvsCCTrainCfg* cfg, void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes,
                             void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AXVS_H */
