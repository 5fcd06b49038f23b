// Host side of the cross-clip tracking module's training tier (included by axvs_train.hip inside namespace axvs::{anonymous}: it
// shares Ctx / pass_fwd / pass_bwd / the GEMM wrappers with the within-clip layer).  Kernels: axvs_cc_train.h.
//
// Forward  = CrossClipTrackingModule.forward in train() mode (CC:275-322): per layer the trajectory attention layer (CC:133-173, post-norm,
//            no positional term, dropout 0, attention-map dropout p_attn_drop), the ASPP (CC:176-201) + residual + LayerNorm (CC:293-295), then
//            the embedding projections and the predictor's training branch (CC:300-309, :45-57) of ALL layers at once (they share
//            weights; BatchNorm statistics stay per layer = per call of the reference).
// Backward = the same chain in reverse; shared head weights accumulate over the layers inside one weight-gradient GEMM.
// Rows of every activation: (b, q, t) -- the reference's clip_query layout [B,Q,Tc,C]; the trajectory attention reads its
// 'b (t q) c' order through the RowMap.

constexpr int kCcC = 256, kCcCm = 128, kCcHeads = 8, kCcMaxLayers = 16;

struct CCShape {
  int B, Q, Tc, V, H, W, K1, nl;
  int rates[3];
  long long M;      // rows per layer: B Q Tc
  long long P;      // pixels per clip: V H W
  size_t E;         // mask logits per layer: B Q Tc P
  Dims d;           // the trajectory pass' view
};

int make_cc_shape(CCShape& s, const AxvsCCTrainCfg* cfg) {
  if (!cfg) return fail(AXVS_ERR_ARG, "null configuration");
  const bool chain = cfg->chain_only != 0;
  if (cfg->B <= 0 || cfg->Q <= 0 || cfg->Tc <= 0 || cfg->num_layers <= 0 || (!chain && (cfg->V <= 0 || cfg->H <= 0 || cfg->W <= 0 || cfg->K1 <= 0)))
    return fail(AXVS_ERR_ARG, "non-positive dimension");
  if (cfg->num_layers > kCcMaxLayers) return fail(AXVS_ERR_ARG, "cross-clip training: num_layers=%d > %d", cfg->num_layers, kCcMaxLayers);
  if (cfg->Tc > 16) return fail(AXVS_ERR_ARG, "cross-clip training: Tc=%d > 16 clips not built", cfg->Tc);
  if (!chain && cfg->Q % 8) return fail(AXVS_ERR_ARG, "cross-clip training: Q=%d must be a multiple of 8", cfg->Q);
  const long long P = chain ? 4 : (long long)cfg->V * cfg->H * cfg->W;       // any pixel count: the einsum GEMMs take unaligned rows and tails
  if ((long long)cfg->B * cfg->Tc > 1024) return fail(AXVS_ERR_ARG, "cross-clip training: B*Tc > 1024");
  for (int k = 0; k < 3; ++k)
    if (cfg->rates[k] <= 0) return fail(AXVS_ERR_ARG, "cross-clip training: atrous rate %d", cfg->rates[k]);
  if (!(cfg->p_attn_drop >= 0.f && cfg->p_attn_drop < 1.f) || !(cfg->p_aspp_drop >= 0.f && cfg->p_aspp_drop < 1.f))
    return fail(AXVS_ERR_ARG, "dropout probability outside [0, 1)");
  s.B = cfg->B; s.Q = cfg->Q; s.Tc = cfg->Tc; s.V = chain ? 1 : cfg->V; s.H = chain ? 1 : cfg->H; s.W = chain ? 4 : cfg->W;
  s.K1 = chain ? 4 : cfg->K1; s.nl = cfg->num_layers;
  for (int k = 0; k < 3; ++k) s.rates[k] = cfg->rates[k];
  s.M = (long long)s.B * s.Q * s.Tc;
  s.P = P;
  s.E = (size_t)s.M * (size_t)P;
  if ((long long)s.M * s.Tc > INT32_MAX || (long long)s.nl * s.M > INT32_MAX) return fail(AXVS_ERR_ARG, "cross-clip training: too many rows");
  // the trajectory pass: T = Tc frames of L = Q tokens; natural rows (b, q, t): frame of row m = m % Tc (HW = 1); the FFN width
  // slot carries the widest weight row of this module (3C: the ASPP convolutions) so the shared scratch is sized for it
  s.d = Dims{s.B, s.Tc, 1, s.Q, kCcC, kCcHeads, 3 * kCcC, kCcC / kCcHeads, s.M, 1};
  return AXVS_OK;
}

struct CCLayerSaved {
  PassSaved ps;
  float *x1, *mean1, *rstd1, *y, *ycat, *p, *meanz, *rstdz, *z, *u, *meanu, *rstdu;
};
struct CCSaved {
  CCLayerSaved l[kCcMaxLayers];
  float *x2, *ce_pre, *me_pre, *ce, *me, *pact, *pooled, *mk_pre, *logits_pre;
  float *mean[4], *rstd[4];   // BatchNorm sites: 0 class projection, 1 mask projection, 2 mask head, 3 pixel space
};

CCSaved carve_cc_saved(Bump& b, const CCShape& s) {
  CCSaved v{};
  const size_t MC = (size_t)s.M * kCcC, G = (size_t)s.nl;
  for (int i = 0; i < s.nl; ++i) {
    CCLayerSaved& l = v.l[i];
    l.ps = carve_pass(b, s.d);
    l.x1 = b.f(MC); l.mean1 = b.f(s.M); l.rstd1 = b.f(s.M);
    l.y = b.f(MC);
    l.ycat = b.f(3 * MC);
    l.p = b.f(MC); l.meanz = b.f(s.M); l.rstdz = b.f(s.M);
    l.z = b.f(MC);
    l.u = b.f(MC); l.meanu = b.f(s.M); l.rstdu = b.f(s.M);
  }
  v.x2 = b.f(G * MC);
  v.ce_pre = b.f(G * MC); v.me_pre = b.f(G * MC); v.ce = b.f(G * MC); v.me = b.f(G * MC);
  v.pact = b.f(G * s.M);
  v.pooled = b.f(G * s.Q * kCcC);
  v.mk_pre = b.f(G * s.M * kCcCm);
  v.logits_pre = b.f(G * s.E);
  const int cs[4] = {kCcC, kCcC, kCcCm, 1};
  for (int k = 0; k < 4; ++k) {
    v.mean[k] = b.f(G * cs[k]);
    v.rstd[k] = b.f(G * cs[k]);
  }
  return v;
}

constexpr int kCcStatBlocks = 256;
struct CCScratch {
  float *xcol, *dxcol, *mk, *kt, *part, *sync, *sums_local;
  // backward
  float *dpre, *dkpart, *dk, *dmk, *dmk_pre, *dme, *dce, *dce_pre, *dme_pre, *dx2h, *dpooled, *part_wa, *part_ba, *part_cls, *dnext[2], *du, *dy, *dz, *dp,
      *dycat, *dx1;
  int dk_ksteps, dk_z;
};

// split-K plan of the mask-kernel gradient (rows x 128 output, contraction over the P pixels of a clip)
void dk_plan(const CCShape& s, int* ksteps, int* z) {
  const int nk = (int)((s.P + kGK - 1) / kGK);
  const long long rows = s.B == 1 ? (long long)s.nl * s.Q : s.Q;
  const int mt = (int)((rows + kGT - 1) / kGT);
  int want = 1024 / (mt > 0 ? mt : 1);         // ~1024 workgroups: several per CU, the k-loop is bound by its load round trips
  want = want < 1 ? 1 : (want > 256 ? 256 : want);
  int ks = (nk + want - 1) / want;
  ks = ks < 4 ? 4 : ks;
  *ksteps = ks;
  *z = (nk + ks - 1) / ks;
}

CCScratch carve_cc_scratch(Bump& b, const CCShape& s, bool backward) {
  CCScratch c{};
  const size_t MC = (size_t)s.M * kCcC, G = (size_t)s.nl;
  c.xcol = b.f(3 * MC);
  c.mk = b.f(G * s.M * kCcCm);
  c.kt = b.f(G * s.M * kCcCm);
  c.part = b.f((size_t)G * kCcStatBlocks * 2 * kCcC);
  c.sync = b.f(G * 4 * kCcC + 8);
  c.sums_local = b.f(G * 4 * kCcC + 8);
  if (!backward) return c;
  dk_plan(s, &c.dk_ksteps, &c.dk_z);
  const size_t rows = s.B == 1 ? G * s.Q : (size_t)s.Q;
  c.dxcol = b.f(3 * MC);
  c.dpre = b.f(G * s.E);
  c.dkpart = b.f((size_t)c.dk_z * rows * kCcCm);
  c.dk = b.f(G * s.M * kCcCm);
  c.dmk = b.f(G * s.M * kCcCm);
  c.dmk_pre = b.f(G * s.M * kCcCm);
  c.dme = b.f(G * MC); c.dce = b.f(G * MC); c.dce_pre = b.f(G * MC); c.dme_pre = b.f(G * MC); c.dx2h = b.f(G * MC);
  c.dpooled = b.f(G * s.Q * kCcC);
  c.part_wa = b.f(G * s.Q * kCcC);
  c.part_ba = b.f(G * s.Q);
  c.part_cls = b.f((size_t)16 * s.K1 * (kCcC + 1));
  c.dnext[0] = b.f(MC); c.dnext[1] = b.f(MC);
  c.du = b.f(MC); c.dy = b.f(MC); c.dz = b.f(MC); c.dp = b.f(MC);
  c.dycat = b.f(3 * MC);
  c.dx1 = b.f(MC);
  return c;
}

struct CCCtx {
  Ctx c;              // the trajectory pass' context (scratch of pass_fwd / pass_bwd, GEMM wrappers, stream)
  CCShape s;
  CCScratch x;
  const AxvsCCTrainCfg* cfg;
  RowMap rm;
  hipStream_t st;
};

inline unsigned eblocks(size_t n) { return blocks(n, 256); }

int cc_sync(const CCCtx& k, float* buf, size_t n) {
  if (!k.cfg->allreduce) return AXVS_OK;
  if (int rc = status()) return rc;
  if (k.cfg->allreduce(k.cfg->allreduce_user, buf, n, k.st) != 0) return fail(AXVS_ERR_LAUNCH, "the caller's all-reduce reported an error");
  return AXVS_OK;
}

// column statistics of x [G][R][C] (shifted by `shift`) -> sums [G][2][C]
// (count_ptr: where the row count goes, written by the same launch)
void cc_bn_stats(const CCCtx& k, const float* x, const float* shift, float* sums, long long R, int C, float* count_ptr = nullptr) {
  int nblk = (int)((R + 31) / 32);
  nblk = nblk > kCcStatBlocks ? kCcStatBlocks : nblk;
  const int rpb = (int)((R + nblk - 1) / nblk);
  nblk = (int)((R + rpb - 1) / rpb);
  hipLaunchKernelGGL(cct_bn_stats_kernel, dim3(nblk, k.s.nl), dim3(256), 0, k.st, x, shift, k.x.part, R, C, rpb);
  hipLaunchKernelGGL(cct_reduce_groups_kernel, dim3((2 * C + 3) / 4, k.s.nl), dim3(256), 0, k.st, (const float*)k.x.part, nblk, 2 * C, sums,
                     (float*)nullptr, count_ptr, (float)R);
}
// sums: this rank's own sums (parameter gradients); sync_copy: the copy that goes through the all-reduce
void cc_bn_bwd_stats(const CCCtx& k, const float* dy, const float* x, const float* mean, const float* rstd, const AxvsBN& bn, float* sums,
                     float* sync_copy, float* count_ptr, long long R, int C, int gelu) {
  int nblk = (int)((R + 31) / 32);
  nblk = nblk > kCcStatBlocks ? kCcStatBlocks : nblk;
  const int rpb = (int)((R + nblk - 1) / nblk);
  nblk = (int)((R + rpb - 1) / rpb);
  hipLaunchKernelGGL(cct_bn_bwd_stats_kernel, dim3(nblk, k.s.nl), dim3(256), 0, k.st, dy, x, mean, rstd, bn.w, bn.b, k.x.part, R, C, rpb, gelu);
  hipLaunchKernelGGL(cct_reduce_groups_kernel, dim3((2 * C + 3) / 4, k.s.nl), dim3(256), 0, k.st, (const float*)k.x.part, nblk, 2 * C, sums,
                     sync_copy, count_ptr, (float)R);
}
// statistics over the E elements of each layer's mask logits: sums [G][2]
void cc_scalar_stats(const CCCtx& k, const float* x, const float* dy, const float* shift, const float* mean, const float* rstd, float* sums,
                     float* sync_copy, float* count_ptr) {
  const size_t per = 16384;
  int nblk = (int)((k.s.E + per - 1) / per);
  nblk = nblk > 4 * kCcStatBlocks ? 4 * kCcStatBlocks : nblk;
  const size_t pb = ((k.s.E + nblk - 1) / nblk + 3) / 4 * 4;
  nblk = (int)((k.s.E + pb - 1) / pb);
  hipLaunchKernelGGL(cct_scalar_stats_kernel, dim3(nblk, k.s.nl), dim3(256), 0, k.st, x, dy, shift, mean, rstd, k.x.part, k.s.E, pb);
  hipLaunchKernelGGL(cct_reduce_groups_kernel, dim3(1, k.s.nl), dim3(256), 0, k.st, (const float*)k.x.part, nblk, 2, sums, sync_copy, count_ptr,
                     (float)k.s.E);
}

const float kVoidBias = logf(0.9f / 0.1f);   // add_bias_towards_void: log((K1 - 1) * 0.9 / 0.1) = log(K1 - 1) + this

// the layer chain (trajectory attention layer -> ASPP -> norms) of all layers: sv.x2[l] = the clip queries after layer l
int cc_chain_forward(const CCCtx& k, const float* cq, const AxvsCCLayerParams* layers, const CCSaved& sv) {
  const CCShape& s = k.s;
  const Ctx& c = k.c;
  const long long M = s.M;
  const int C = kCcC, G = s.nl;
  const size_t MC = (size_t)M * C;
  const bool ex = g_train_exact != 0;
  const Drop none = make_drop(0.f, 0, 0);
  int rc;
  for (int l = 0; l < G; ++l) {
    const CCLayerSaved& L = sv.l[l];
    const AxvsCCLayerParams& p = layers[l];
    const float* xin = l ? sv.x2 + (size_t)(l - 1) * MC : cq;
    // trajectory attention layer, post-norm (CC:155-161)
    if ((rc = pass_fwd(c, xin, nullptr, L.x1, p.attn, L.ps, k.rm, s.B, make_drop(k.cfg->p_attn_drop, k.cfg->seed, 10 + 2 * l), none)) != AXVS_OK) return rc;
    hipLaunchKernelGGL(tr_ln_fwd_kernel, dim3(blocks(M, 4)), dim3(256), 0, k.st, (const float*)L.x1, p.norm_w, p.norm_b, L.y, L.mean1, L.rstd1, M, C, 1e-5f);
    // ASPP: three dilated convolutions over the clip axis as GEMMs over the im2col, written side by side (torch.cat, CC:197)
    for (int b = 0; b < 3; ++b) {
      hipLaunchKernelGGL(cct_im2col_kernel, dim3(eblocks(3 * MC)), dim3(256), 0, k.st, (const float*)L.y, k.x.xcol, M, s.Tc, C, s.rates[b]);
      const GemmEpi e{p.aspp_b[b], 1.f, 0, none, 0.f};
      if ((rc = c.g.nt(k.x.xcol, p.aspp_w[b], L.ycat + b * C, M, C, 3 * C, GemmLd{3 * C, 3 * C, 3 * C, 0}, e, ex)) != AXVS_OK) return rc;
    }
    if ((rc = c.g.fwd(L.ycat, p.aspp_proj_w, L.p, M, C, 3 * C, 0.f, nullptr, ex)) != AXVS_OK) return rc;
    hipLaunchKernelGGL(tr_ln_fwd_kernel, dim3(blocks(M, 4)), dim3(256), 0, k.st, (const float*)L.p, p.aspp_norm_w, p.aspp_norm_b, L.z, L.meanz, L.rstdz, M, C, 1e-6f);
    hipLaunchKernelGGL(cct_gelu_drop_res_kernel, dim3(eblocks(MC)), dim3(256), 0, k.st, (const float*)L.z, (const float*)L.y, L.u, M, s.Tc, C,
                       make_drop(k.cfg->p_aspp_drop, k.cfg->seed, 11 + 2 * l));
    hipLaunchKernelGGL(tr_ln_fwd_kernel, dim3(blocks(M, 4)), dim3(256), 0, k.st, (const float*)L.u, p.conv_norm_w, p.conv_norm_b, sv.x2 + (size_t)l * MC,
                       L.meanu, L.rstdu, M, C, 1e-5f);
  }
  return status();
}

int cc_forward(const CCCtx& k, const float* cq, const float* pf, float* logits_out, float* masks_out, float* bn_stats_out,
               const AxvsCCLayerParams* layers, const AxvsCCHeadParams& hp, const CCSaved& sv) {
  const CCShape& s = k.s;
  const Ctx& c = k.c;
  const long long M = s.M;
  const int C = kCcC, G = s.nl;
  const bool ex = g_train_exact != 0;
  int rc;
  if ((rc = cc_chain_forward(k, cq, layers, sv)) != AXVS_OK) return rc;
  // ---- heads of all layers ----
  const long long GM = (long long)G * M;
  float* const sync = k.x.sync;
  if ((rc = c.g.fwd(sv.x2, hp.class_proj_w, sv.ce_pre, GM, C, C, 0.f, nullptr, ex)) != AXVS_OK) return rc;
  if ((rc = c.g.fwd(sv.x2, hp.mask_proj_w, sv.me_pre, GM, C, C, 0.f, nullptr, ex)) != AXVS_OK) return rc;
  cc_bn_stats(k, sv.ce_pre, hp.class_proj_bn.mean, sync, M, C);
  cc_bn_stats(k, sv.me_pre, hp.mask_proj_bn.mean, sync + (size_t)G * 2 * C, M, C, sync + (size_t)G * 4 * C);
  if ((rc = cc_sync(k, sync, (size_t)G * 4 * C + 1)) != AXVS_OK) return rc;
  float* so = bn_stats_out;                         // [class_proj [G][2][C] | mask_proj [G][2][C] | mask_head [G][2][Cm] | pixel [G][2]]
  hipLaunchKernelGGL(cct_bn_finalize_kernel, dim3(1, G), dim3(256), 0, k.st, (const float*)sync, (const float*)(sync + (size_t)G * 4 * C),
                     hp.class_proj_bn.mean, 1e-3f, sv.mean[0], sv.rstd[0], so, C);
  hipLaunchKernelGGL(cct_bn_finalize_kernel, dim3(1, G), dim3(256), 0, k.st, (const float*)(sync + (size_t)G * 2 * C),
                     (const float*)(sync + (size_t)G * 4 * C), hp.mask_proj_bn.mean, 1e-3f, sv.mean[1], sv.rstd[1], so + (size_t)G * 2 * C, C);
  hipLaunchKernelGGL(cct_bn_apply_kernel, dim3(eblocks((size_t)GM * C)), dim3(256), 0, k.st, (const float*)sv.ce_pre, (const float*)sv.mean[0],
                     (const float*)sv.rstd[0], hp.class_proj_bn.w, hp.class_proj_bn.b, sv.ce, M, C, G, 1);
  hipLaunchKernelGGL(cct_bn_apply_kernel, dim3(eblocks((size_t)GM * C)), dim3(256), 0, k.st, (const float*)sv.me_pre, (const float*)sv.mean[1],
                     (const float*)sv.rstd[1], hp.mask_proj_bn.w, hp.mask_proj_bn.b, sv.me, M, C, G, 1);
  // class branch (CC:48-52)
  hipLaunchKernelGGL(cct_act_pool_fwd_kernel, dim3(s.Q, G), dim3(256), 0, k.st, (const float*)sv.ce, hp.act_head_w, hp.act_head_b, sv.pact, sv.pooled, s.B,
                     s.Q, s.Tc, C);
  hipLaunchKernelGGL(cct_small_linear_fwd_kernel, dim3(blocks((size_t)G * s.Q * s.K1, 4)), dim3(256), 0, k.st, (const float*)sv.pooled, hp.class_head_w,
                     hp.class_head_b, logits_out, G * s.Q, C, s.K1, logf((float)(s.K1 - 1)) + kVoidBias);
  // mask branch (CC:53-57)
  if ((rc = c.g.fwd(sv.me, hp.mask_head_w, sv.mk_pre, GM, kCcCm, C, 0.f, nullptr, ex)) != AXVS_OK) return rc;
  cc_bn_stats(k, sv.mk_pre, hp.mask_head_bn.mean, sync, M, kCcCm, sync + (size_t)G * 2 * kCcCm);
  if ((rc = cc_sync(k, sync, (size_t)G * 2 * kCcCm + 1)) != AXVS_OK) return rc;
  so += (size_t)G * 4 * C;
  hipLaunchKernelGGL(cct_bn_finalize_kernel, dim3(1, G), dim3(256), 0, k.st, (const float*)sync, (const float*)(sync + (size_t)G * 2 * kCcCm),
                     hp.mask_head_bn.mean, 1e-3f, sv.mean[2], sv.rstd[2], so, kCcCm);
  hipLaunchKernelGGL(cct_bn_apply_kernel, dim3(eblocks((size_t)GM * kCcCm)), dim3(256), 0, k.st, (const float*)sv.mk_pre, (const float*)sv.mean[2],
                     (const float*)sv.rstd[2], hp.mask_head_bn.w, hp.mask_head_bn.b, k.x.mk, M, kCcCm, G, 0);
  hipLaunchKernelGGL(cct_kern_pack_kernel, dim3(eblocks((size_t)GM * kCcCm)), dim3(256), 0, k.st, (const float*)k.x.mk, k.x.kt, G, s.B, s.Q, s.Tc, kCcCm);
  // 'bchw,bcn->bnhw' per clip, all layers' kernels at once when the output rows (layer, query) are evenly strided (B == 1)
  const long long TP = (long long)s.Tc * s.P;
  const int GQ = G * s.Q;
  const int al_pf = row_align(pf, s.P, s.P), al_lg = row_align(sv.logits_pre, s.P, s.P);   // rows start at multiples of P floats
  // B == 1 and whole 128-row tiles per layer (the shipped 128 queries): the einsum's epilogue also leaves the tile sums the
  // one-channel BatchNorm needs -- one pass over the 0.8 GB of a layer's mask logits saved
  const int tiles_p = (int)((s.P + kGT - 1) / kGT), qtiles = s.Q / kGT;
  const long long stat_nblk = (long long)s.Tc * tiles_p * qtiles;
  const bool fused_stats = s.B == 1 && s.Q % kGT == 0 && stat_nblk <= (long long)kCcStatBlocks * kCcC;
  if (s.B == 1) {
    for (int t = 0; t < s.Tc; ++t) {
      GemmLd stat{0, 0, 0, 0};
      stat.stat_part = k.x.part; stat.stat_shift = hp.pixel_bn.mean; stat.stat_nblk = (int)stat_nblk; stat.stat_blk0 = t * tiles_p * qtiles;
      stat.stat_rows = s.Q;
      if ((rc = c.g.tn_direct(k.x.kt + (size_t)t * kCcCm * GQ, pf + (size_t)t * s.P, sv.logits_pre + (size_t)t * s.P, kCcCm, GQ, (int)s.P, GQ, TP, TP,
                              al_pf, al_lg, fused_stats ? &stat : nullptr)) != AXVS_OK)
        return rc;
    }
  } else {
    for (int g = 0; g < G; ++g)
      for (int b = 0; b < s.B; ++b)
        for (int t = 0; t < s.Tc; ++t)
          if ((rc = c.g.tn_direct(k.x.kt + ((size_t)b * s.Tc + t) * kCcCm * GQ + (size_t)g * s.Q, pf + (size_t)b * kCcCm * TP + (size_t)t * s.P,
                                  sv.logits_pre + (((size_t)g * s.B + b) * s.Q) * TP + (size_t)t * s.P, kCcCm, s.Q, (int)s.P, GQ, TP, TP, al_pf,
                                  al_lg)) != AXVS_OK)
            return rc;
  }
  // one-channel BatchNorm over each layer's mask logits (CC:56)
  if (fused_stats)
    hipLaunchKernelGGL(cct_reduce_groups_kernel, dim3(1, G), dim3(256), 0, k.st, (const float*)k.x.part, (int)stat_nblk, 2, sync, (float*)nullptr,
                       sync + (size_t)G * 2, (float)s.E);
  else cc_scalar_stats(k, sv.logits_pre, nullptr, hp.pixel_bn.mean, nullptr, nullptr, sync, nullptr, sync + (size_t)G * 2);
  if ((rc = cc_sync(k, sync, (size_t)G * 2 + 1)) != AXVS_OK) return rc;
  so += (size_t)G * 2 * kCcCm;
  hipLaunchKernelGGL(cct_bn_finalize_kernel, dim3(1, G), dim3(256), 0, k.st, (const float*)sync, (const float*)(sync + (size_t)G * 2), hp.pixel_bn.mean,
                     1e-3f, sv.mean[3], sv.rstd[3], so, 1);
  hipLaunchKernelGGL(cct_scalar_bn_apply_kernel, dim3(eblocks(s.E / 4), G), dim3(256), 0, k.st, (const float*)sv.logits_pre, (const float*)sv.mean[3],
                     (const float*)sv.rstd[3], hp.pixel_bn.w, hp.pixel_bn.b, masks_out, s.E / 4);
  return status();
}

int cc_chain_backward(const CCCtx& k, const float* cq, const AxvsCCLayerParams* layers, const AxvsCCLayerGrads* lg, float* d_cq, const CCSaved& sv);

int cc_backward(const CCCtx& k, const float* d_logits, const float* d_masks, const float* cq, const float* pf, const AxvsCCLayerParams* layers,
                const AxvsCCHeadParams& hp, const AxvsCCLayerGrads* lg, const AxvsCCHeadGrads& hg, float* d_cq, const CCSaved& sv) {
  const CCShape& s = k.s;
  const Ctx& c = k.c;
  const CCScratch& x = k.x;
  const long long M = s.M;
  const int C = kCcC, G = s.nl, Cm = kCcCm;
  const size_t MC = (size_t)M * C;
  const long long GM = (long long)G * M;
  const long long TP = (long long)s.Tc * s.P;
  const int GQ = G * s.Q;
  float* const sync = x.sync;
  float* const loc = x.sums_local;
  int rc;
  // ---- pixel-space BatchNorm ----
  cc_scalar_stats(k, sv.logits_pre, d_masks, nullptr, sv.mean[3], sv.rstd[3], loc, sync, sync + (size_t)G * 2);
  hipLaunchKernelGGL(cct_bn_param_grads_kernel, dim3(1), dim3(256), 0, k.st, (const float*)loc, hg.pixel_bn.w, hg.pixel_bn.b, 1, G);
  if ((rc = cc_sync(k, sync, (size_t)G * 2 + 1)) != AXVS_OK) return rc;
  // ---- mask kernels: dk[(b t)][(g q)][c] = sum_p dpre[g][b][q][t P + p] pf[b][c][t P + p] (split-K partials, summed in a fixed order) ----
  const GemmEpi plain{nullptr, 1.f, 0, make_drop(0.f, 0, 0), 0.f};
  GemmLd ldk{TP, TP, Cm, x.dk_ksteps};
  ldk.al_b = row_align(pf, s.P, s.P);
  if (s.B == 1) {
    // B == 1: the BatchNorm's input gradient dpre = c1 d_masks + c2 logits_pre + c3 (per layer) is formed in the loader of the GEMM
    // that contracts it with the pixel features -- no pass that writes 0.8 GB per layer and reads it back
    float* const coef = x.sums_local + (size_t)G * 2;          // (behind the local sums: 3 floats per layer)
    hipLaunchKernelGGL(cct_scalar_bn_bwd_coef_kernel, dim3(1), dim3(64), 0, k.st, (const float*)sv.mean[3], (const float*)sv.rstd[3], hp.pixel_bn.w,
                       (const float*)sync, (const float*)(sync + (size_t)G * 2), coef, G);
    const int al_d = row_align(d_masks, s.P, s.P), al_l = row_align(sv.logits_pre, s.P, s.P);
    ldk.al_a = al_d < al_l ? al_d : al_l;
    ldk.aff = coef;
    ldk.aff_rows = s.Q;
    for (int t = 0; t < s.Tc; ++t) {
      ldk.a2 = sv.logits_pre + (size_t)t * s.P;
      if ((rc = c.g.nt(d_masks + (size_t)t * s.P, pf + (size_t)t * s.P, x.dkpart, GQ, Cm, (int)s.P, ldk, plain, false, x.dk_z)) != AXVS_OK) return rc;
      const size_t n = (size_t)GQ * Cm;
      hipLaunchKernelGGL(tr_colsum_final_kernel, dim3(blocks(n, 256)), dim3(256), 0, k.st, (const float*)x.dkpart, x.dk_z, n, x.dk + (size_t)t * n);
    }
  } else {
    hipLaunchKernelGGL(cct_scalar_bn_bwd_apply_kernel, dim3(eblocks(s.E / 4), G), dim3(256), 0, k.st, d_masks, (const float*)sv.logits_pre,
                       (const float*)sv.mean[3], (const float*)sv.rstd[3], hp.pixel_bn.w, (const float*)sync, (const float*)(sync + (size_t)G * 2),
                       x.dpre, s.E / 4);
    ldk.al_a = row_align(x.dpre, s.P, s.P);
    for (int g = 0; g < G; ++g)
      for (int b = 0; b < s.B; ++b)
        for (int t = 0; t < s.Tc; ++t) {
          if ((rc = c.g.nt(x.dpre + (((size_t)g * s.B + b) * s.Q) * TP + (size_t)t * s.P, pf + (size_t)b * Cm * TP + (size_t)t * s.P, x.dkpart, s.Q, Cm,
                           (int)s.P, ldk, plain, false, x.dk_z)) != AXVS_OK)
            return rc;
          const size_t n = (size_t)s.Q * Cm;
          hipLaunchKernelGGL(tr_colsum_final_kernel, dim3(blocks(n, 256)), dim3(256), 0, k.st, (const float*)x.dkpart, x.dk_z, n,
                             x.dk + (((size_t)b * s.Tc + t) * GQ + (size_t)g * s.Q) * Cm);
        }
  }
  hipLaunchKernelGGL(cct_kern_unpack_kernel, dim3(eblocks((size_t)GM * Cm)), dim3(256), 0, k.st, (const float*)x.dk, x.dmk, G, s.B, s.Q, s.Tc, Cm);
  // ---- mask head: conv 1x1 + BatchNorm (no activation) ----
  cc_bn_bwd_stats(k, x.dmk, sv.mk_pre, sv.mean[2], sv.rstd[2], hp.mask_head_bn, loc, sync, sync + (size_t)G * 2 * Cm, M, Cm, 0);
  hipLaunchKernelGGL(cct_bn_param_grads_kernel, dim3(1), dim3(256), 0, k.st, (const float*)loc, hg.mask_head_bn.w, hg.mask_head_bn.b, Cm, G);
  if ((rc = cc_sync(k, sync, (size_t)G * 2 * Cm + 1)) != AXVS_OK) return rc;
  hipLaunchKernelGGL(cct_bn_bwd_apply_kernel, dim3(eblocks((size_t)GM * Cm)), dim3(256), 0, k.st, (const float*)x.dmk, (const float*)sv.mk_pre,
                     (const float*)sv.mean[2], (const float*)sv.rstd[2], hp.mask_head_bn.w, hp.mask_head_bn.b, (const float*)sync,
                     (const float*)(sync + (size_t)G * 2 * Cm), x.dmk_pre, M, Cm, G, 0);
  if ((rc = c.wgrad(x.dmk_pre, sv.me, hg.mask_head_w, GM, Cm, C)) != AXVS_OK) return rc;
  if ((rc = c.dgrad(x.dmk_pre, hp.mask_head_w, x.dme, GM, Cm, C, 0.f)) != AXVS_OK) return rc;
  // ---- class head, activation pooling ----
  constexpr int kWSplits = 16;
  hipLaunchKernelGGL(cct_small_linear_bwd_w_kernel, dim3(s.K1, kWSplits), dim3(256), 0, k.st, d_logits, (const float*)sv.pooled, x.part_cls, GQ, C, s.K1);
  hipLaunchKernelGGL(cct_small_linear_bwd_w_final_kernel, dim3(s.K1), dim3(256), 0, k.st, (const float*)x.part_cls, hg.class_head_w, hg.class_head_b,
                     kWSplits, C, s.K1);
  hipLaunchKernelGGL(cct_small_linear_bwd_x_kernel, dim3(GQ), dim3(256), 0, k.st, d_logits, hp.class_head_w, x.dpooled, GQ, C, s.K1);
  hipLaunchKernelGGL(cct_act_pool_bwd_kernel, dim3(s.Q, G), dim3(256), 0, k.st, (const float*)sv.ce, hp.act_head_w, (const float*)sv.pact,
                     (const float*)x.dpooled, x.dce, x.part_wa, x.part_ba, s.B, s.Q, s.Tc, C);
  hipLaunchKernelGGL(tr_colsum_final_kernel, dim3(blocks(C, 256)), dim3(256), 0, k.st, (const float*)x.part_wa, GQ, (size_t)C, hg.act_head_w);
  hipLaunchKernelGGL(cct_reduce_groups_kernel, dim3(1, 1), dim3(256), 0, k.st, (const float*)x.part_ba, GQ, 1, hg.act_head_b, (float*)nullptr,
                     (float*)nullptr, 0.f);
  // ---- the two embedding projections: conv 1x1 + BatchNorm + GELU ----
  cc_bn_bwd_stats(k, x.dce, sv.ce_pre, sv.mean[0], sv.rstd[0], hp.class_proj_bn, loc, sync, nullptr, M, C, 1);
  cc_bn_bwd_stats(k, x.dme, sv.me_pre, sv.mean[1], sv.rstd[1], hp.mask_proj_bn, loc + (size_t)G * 2 * C, sync + (size_t)G * 2 * C,
                  sync + (size_t)G * 4 * C, M, C, 1);
  hipLaunchKernelGGL(cct_bn_param_grads_kernel, dim3(1), dim3(256), 0, k.st, (const float*)loc, hg.class_proj_bn.w, hg.class_proj_bn.b, C, G);
  hipLaunchKernelGGL(cct_bn_param_grads_kernel, dim3(1), dim3(256), 0, k.st, (const float*)(loc + (size_t)G * 2 * C), hg.mask_proj_bn.w, hg.mask_proj_bn.b, C, G);
  if ((rc = cc_sync(k, sync, (size_t)G * 4 * C + 1)) != AXVS_OK) return rc;
  hipLaunchKernelGGL(cct_bn_bwd_apply_kernel, dim3(eblocks((size_t)GM * C)), dim3(256), 0, k.st, (const float*)x.dce, (const float*)sv.ce_pre,
                     (const float*)sv.mean[0], (const float*)sv.rstd[0], hp.class_proj_bn.w, hp.class_proj_bn.b, (const float*)sync,
                     (const float*)(sync + (size_t)G * 4 * C), x.dce_pre, M, C, G, 1);
  hipLaunchKernelGGL(cct_bn_bwd_apply_kernel, dim3(eblocks((size_t)GM * C)), dim3(256), 0, k.st, (const float*)x.dme, (const float*)sv.me_pre,
                     (const float*)sv.mean[1], (const float*)sv.rstd[1], hp.mask_proj_bn.w, hp.mask_proj_bn.b, (const float*)(sync + (size_t)G * 2 * C),
                     (const float*)(sync + (size_t)G * 4 * C), x.dme_pre, M, C, G, 1);
  if ((rc = c.wgrad(x.dce_pre, sv.x2, hg.class_proj_w, GM, C, C)) != AXVS_OK) return rc;
  if ((rc = c.wgrad(x.dme_pre, sv.x2, hg.mask_proj_w, GM, C, C)) != AXVS_OK) return rc;
  // (three-piece products: the rows of d*_pre sum to zero per channel -- BatchNorm removes a constant -- so the last layer's
  //  conv_norms bias gradient is a sum that cancels exactly, and two-piece rounding would be all that is left of it)
  if ((rc = c.dgrad(x.dce_pre, hp.class_proj_w, x.dx2h, GM, C, C, 0.f, 0, true)) != AXVS_OK) return rc;
  if ((rc = c.dgrad(x.dme_pre, hp.mask_proj_w, x.dx2h, GM, C, C, 1.f, 0, true)) != AXVS_OK) return rc;
  return cc_chain_backward(k, cq, layers, lg, d_cq, sv);
}

// the layer chain, last layer first; x.dx2h [nl][M][C] holds the gradient that reaches each layer's output from OUTSIDE the chain
// (the heads), the next layer's input gradient is added on the way
int cc_chain_backward(const CCCtx& k, const float* cq, const AxvsCCLayerParams* layers, const AxvsCCLayerGrads* lg, float* d_cq, const CCSaved& sv) {
  const CCShape& s = k.s;
  const Ctx& c = k.c;
  const CCScratch& x = k.x;
  const long long M = s.M;
  const int C = kCcC, G = s.nl;
  const size_t MC = (size_t)M * C;
  int rc;
  const Drop none = make_drop(0.f, 0, 0);
  for (int l = G - 1; l >= 0; --l) {
    const CCLayerSaved& L = sv.l[l];
    const AxvsCCLayerParams& p = layers[l];
    const AxvsCCLayerGrads& g = lg[l];
    float* const dx2 = x.dx2h + (size_t)l * MC;                       // heads' share; + the next layer's input gradient
    if (l < G - 1) hipLaunchKernelGGL(cct_add_inplace_kernel, dim3(eblocks(MC)), dim3(256), 0, k.st, dx2, (const float*)x.dnext[(l + 1) & 1], MC);
    // conv_norms[l] on u = y + dropout(gelu(z))       (CC:293-295)
    c.colsum(dx2, M, C, g.conv_norm_b, L.u, L.meanu, L.rstdu, g.conv_norm_w);
    hipLaunchKernelGGL(tr_ln_bwd_kernel, dim3(blocks(M, 4)), dim3(256), 0, k.st, (const float*)dx2, (const float*)L.u, p.conv_norm_w, (const float*)L.meanu,
                       (const float*)L.rstdu, x.du, M, C);
    hipLaunchKernelGGL(cct_gelu_drop_bwd_kernel, dim3(eblocks(MC)), dim3(256), 0, k.st, (const float*)x.du, (const float*)L.z, x.dz, x.dy, M, s.Tc, C,
                       make_drop(k.cfg->p_aspp_drop, k.cfg->seed, 11 + 2 * l));
    // channels-first LayerNorm of the ASPP projection, the projection, the three convolutions
    c.colsum(x.dz, M, C, g.aspp_norm_b, L.p, L.meanz, L.rstdz, g.aspp_norm_w);
    hipLaunchKernelGGL(tr_ln_bwd_kernel, dim3(blocks(M, 4)), dim3(256), 0, k.st, (const float*)x.dz, (const float*)L.p, p.aspp_norm_w, (const float*)L.meanz,
                       (const float*)L.rstdz, x.dp, M, C);
    if ((rc = c.wgrad(x.dp, L.ycat, g.aspp_proj_w, M, C, 3 * C)) != AXVS_OK) return rc;
    if ((rc = c.dgrad(x.dp, p.aspp_proj_w, x.dycat, M, C, 3 * C, 0.f)) != AXVS_OK) return rc;
    for (int b = 0; b < 3; ++b) {
      hipLaunchKernelGGL(cct_im2col_kernel, dim3(eblocks(3 * MC)), dim3(256), 0, k.st, (const float*)L.y, x.xcol, M, s.Tc, C, s.rates[b]);
      if ((rc = c.wgrad(x.dycat + b * C, x.xcol, g.aspp_w[b], M, C, 3 * C, g.aspp_b[b], 3 * C, 0)) != AXVS_OK) return rc;
      if ((rc = c.dgrad(x.dycat + b * C, p.aspp_w[b], x.dxcol, M, C, 3 * C, 0.f, 3 * C)) != AXVS_OK) return rc;
      hipLaunchKernelGGL(cct_col2im_add_kernel, dim3(eblocks(MC)), dim3(256), 0, k.st, (const float*)x.dxcol, x.dy, M, s.Tc, C, s.rates[b]);
    }
    // the trajectory layer's post-norm, then the attention itself
    c.colsum(x.dy, M, C, g.norm_b, L.x1, L.mean1, L.rstd1, g.norm_w);
    hipLaunchKernelGGL(tr_ln_bwd_kernel, dim3(blocks(M, 4)), dim3(256), 0, k.st, (const float*)x.dy, (const float*)L.x1, p.norm_w, (const float*)L.mean1,
                       (const float*)L.rstd1, x.dx1, M, C);
    const float* xin = l ? sv.x2 + (size_t)(l - 1) * MC : cq;
    float* const d_in = l ? x.dnext[l & 1] : d_cq;
    if ((rc = pass_bwd(c, x.dx1, xin, nullptr, p.attn, g.attn, L.ps, k.rm, s.B, make_drop(k.cfg->p_attn_drop, k.cfg->seed, 10 + 2 * l), none, d_in, nullptr,
                       false)) != AXVS_OK)
      return rc;
  }
  return status();
}

int cc_check_ptrs(const void* p, size_t bytes, const char* what) {
  const void* const* f = static_cast<const void* const*>(p);
  for (size_t i = 0; i < bytes / sizeof(void*); ++i)
    if (!f[i]) return fail(AXVS_ERR_ARG, "null pointer (field %zu of %s)", i, what);
  return AXVS_OK;
}

int cc_setup(CCCtx& k, const AxvsCCTrainCfg* cfg, void* scratch, size_t scratch_bytes, void* saved, size_t saved_bytes, CCSaved& sv, bool backward,
             void* stream) {
  int rc;
  if ((rc = make_cc_shape(k.s, cfg)) != AXVS_OK) return rc;
  k.cfg = cfg;
  Bump sb(saved), cb(scratch);
  sv = carve_cc_saved(sb, k.s);
  k.c.d = k.s.d;
  k.c.sc = carve_scratch(cb, k.s.d, backward);
  k.x = carve_cc_scratch(cb, k.s, backward);
  if (sb.off > saved_bytes || cb.off > scratch_bytes)
    return fail(AXVS_ERR_WORKSPACE, "training buffers too small: saved %zu < %zu or scratch %zu < %zu", saved_bytes, sb.off, scratch_bytes, cb.off);
  k.st = k.c.st = static_cast<hipStream_t>(stream);
  k.c.scale = 1.f / sqrtf((float)k.s.d.D);
  // tokens of a sequence in the reference's 'b (t q) c' order (CC:284) over natural rows (b, q, t)
  k.rm = RowMap{k.s.Tc * k.s.Q, k.s.Q, 1, (long long)k.s.Q * k.s.Tc, 1, k.s.Tc, 0};
  return k.c.g.init(k.st);
}
