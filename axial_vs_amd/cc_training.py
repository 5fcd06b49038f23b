"""Training path of CrossClipTrackingModule: autograd over libaxvs.so's cross-clip training tier.

Reference: ``CrossClipTrackingModule.forward`` in ``train()`` mode under autograd, CC = MaXTron_Video-kMaX/maxtron_deeplab/modeling/
cross_clip_tracking_module/maxtron_cross_clip_tracking_module.py:275-322 with the predictor's training branch (:45-57) -- the module the
reference trains on its own with backbone and segmentation head frozen (maxtron_cc_model.py:104-108).  Forward and backward both run in
the library (``axvs_cc_module_train_fwd`` / ``_bwd``, include/axvs.h); this file is the ``torch.autograd.Function`` that binds them, the
running-statistics update of the four BatchNorm sites (momentum arithmetic on [C] vectors) and the SyncBatchNorm all-reduce callback.

* (Sync)BatchNorm: in ``train()`` mode the four ``ConvBN(norm='syncbn')`` sites normalise with BATCH statistics.  When
  ``torch.distributed`` is initialised with more than one rank the library's partial sums are summed over the ranks (RCCL
  all-reduce of 2 C + 1 floats per site and layer, three calls per forward and three per backward), as ``nn.SyncBatchNorm`` does.
* dropout masks are the same counter-based hash as the layer's training tier (sites 10 + 2 l: attention maps of layer l, 11 + 2 l:
  ASPP ``_proj_drop``); ``seed`` from torch's CPU generator or ``module.dropout_seed``.
* no gradient is produced for ``panoptic_features`` (the frozen segmenter's pixel features).
"""
from __future__ import annotations

import ctypes as C
from typing import List

import torch
import torch.distributed as dist
from torch import Tensor

from . import _lib
from .training import _f32c

_BN_SITES = ("_class_embedding_projection", "_mask_embedding_projection")


def _bn_modules(mod):
    return [mod._class_embedding_projection.norm, mod._mask_embedding_projection.norm, mod._predictor._transformer_mask_head.norm,
            mod._predictor._pixel_space_mask_batch_norm]


def module_parameters(mod) -> List[Tensor]:
    """Every trainable tensor of the module, in the order the autograd Function returns gradients."""
    ps: List[Tensor] = []
    for i in range(mod.num_layers):
        lay, asp, cn = mod.transformer_trajectory_self_attention_layers[i], mod.conv_short_aggregate_layers[i], mod.conv_norms[i]
        at = lay.self_attn
        ps += [at.qkv.weight, at.qkv.bias, at.proj_q.weight, at.proj_q.bias, at.proj_kv.weight, at.proj_kv.bias, at.proj.weight, at.proj.bias,
               lay.norm.weight, lay.norm.bias]
        for k in range(3):
            conv = getattr(asp, f"_aspp_conv{k}")
            ps += [conv.weight, conv.bias]
        ps += [asp._proj_conv_bn_act.conv.weight, asp._proj_conv_bn_act.norm.weight, asp._proj_conv_bn_act.norm.bias, cn.weight, cn.bias]
    pr = mod._predictor
    ps += [mod._class_embedding_projection.conv.weight, mod._class_embedding_projection.norm.weight, mod._class_embedding_projection.norm.bias,
           mod._mask_embedding_projection.conv.weight, mod._mask_embedding_projection.norm.weight, mod._mask_embedding_projection.norm.bias,
           pr._transformer_mask_head.conv.weight, pr._transformer_mask_head.norm.weight, pr._transformer_mask_head.norm.bias,
           pr._transformer_class_head.conv.weight, pr._transformer_class_head.conv.bias,
           pr._transformer_class_activation_head.conv.weight, pr._transformer_class_activation_head.conv.bias,
           pr._pixel_space_mask_batch_norm.weight, pr._pixel_space_mask_batch_norm.bias]
    return ps


_PER_LAYER = 21
_HEAD = 15


def _layer_struct(ptrs: List[int], Cc: int = 256):
    """AxvsCCLayerParams (or, with gradient buffers, AxvsCCLayerGrads: same layout) from one layer's 21 tensors."""
    s = _lib.AxvsCCLayerParams()
    qkv_w, qkv_b = ptrs[0], ptrs[1]
    s.attn = _lib.AxvsTrajParams(qkv_w, qkv_b, qkv_w + 4 * Cc * Cc, qkv_b + 4 * Cc, qkv_w + 8 * Cc * Cc, qkv_b + 8 * Cc, *ptrs[2:8])
    s.norm_w, s.norm_b = ptrs[8], ptrs[9]
    for k in range(3):
        s.aspp_w[k], s.aspp_b[k] = ptrs[10 + 2 * k], ptrs[11 + 2 * k]
    s.aspp_proj_w, s.aspp_norm_w, s.aspp_norm_b, s.conv_norm_w, s.conv_norm_b = ptrs[16:21]
    return s


def _head_struct(ptrs: List[int], running) -> _lib.AxvsCCHeadParams:
    h = _lib.AxvsCCHeadParams()
    h.class_proj_w, h.class_proj_bn = ptrs[0], _lib.AxvsBN(ptrs[1], ptrs[2], running[0][0], running[0][1])
    h.mask_proj_w, h.mask_proj_bn = ptrs[3], _lib.AxvsBN(ptrs[4], ptrs[5], running[1][0], running[1][1])
    h.mask_head_w, h.mask_head_bn = ptrs[6], _lib.AxvsBN(ptrs[7], ptrs[8], running[2][0], running[2][1])
    h.class_head_w, h.class_head_b, h.act_head_w, h.act_head_b = ptrs[9:13]
    h.pixel_bn = _lib.AxvsBN(ptrs[13], ptrs[14], running[3][0], running[3][1])
    return h


def _head_grads(ptrs: List[int]) -> _lib.AxvsCCHeadGrads:
    h = _lib.AxvsCCHeadGrads()
    h.class_proj_w, h.class_proj_bn = ptrs[0], _lib.AxvsBNGrads(ptrs[1], ptrs[2])
    h.mask_proj_w, h.mask_proj_bn = ptrs[3], _lib.AxvsBNGrads(ptrs[4], ptrs[5])
    h.mask_head_w, h.mask_head_bn = ptrs[6], _lib.AxvsBNGrads(ptrs[7], ptrs[8])
    h.class_head_w, h.class_head_b, h.act_head_w, h.act_head_b = ptrs[9:13]
    h.pixel_bn = _lib.AxvsBNGrads(ptrs[13], ptrs[14])
    return h


class _AllReduce:
    """The library's SyncBatchNorm hook: sums `n` floats at a device address inside `buf` over the ranks of `group`."""

    def __init__(self, buf: Tensor, group):
        self.buf, self.group, self.error = buf, group, None
        self.fn = _lib.ALLREDUCE_FN(self._call)

    def _call(self, user, ptr, n, stream):
        try:
            off = int(ptr) - self.buf.data_ptr()
            if off < 0 or off + 4 * n > self.buf.numel():
                raise RuntimeError("all-reduce buffer outside the scratch tensor")
            dist.all_reduce(self.buf[off:off + 4 * n].view(torch.float32), group=self.group)
            return 0
        except Exception as e:   # an exception may not cross the C frame
            self.error = e
            return 1


def _cfg(dims, rates, p_attn, p_aspp, seed, hook) -> _lib.AxvsCCTrainCfg:
    B, Q, Tc, V, H, W, K1, nl = dims
    c = _lib.AxvsCCTrainCfg()
    c.B, c.Q, c.Tc, c.V, c.H, c.W, c.K1, c.num_layers = B, Q, Tc, V, H, W, K1, nl
    for k in range(3):
        c.rates[k] = int(rates[k])
    c.p_attn_drop, c.p_aspp_drop, c.seed = float(p_attn), float(p_aspp), int(seed)
    c.allreduce = hook.fn if hook is not None else _lib.ALLREDUCE_FN()
    c.allreduce_user = None
    c.chain_only = 0
    return c


def _sync_group():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist.group.WORLD
    return None


class _CCModuleTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, clip_query, panoptic_features, cfg, running, *params):
        from .modules import _stream
        dims, rates, p_attn, p_aspp, seed = cfg
        if not clip_query.is_cuda:
            raise RuntimeError("axial_vs_amd: the training tier needs GPU tensors; there is no CPU fallback")
        B, Q, Tc, V, H, W, K1, nl = dims
        cq, pf = _f32c(clip_query), _f32c(panoptic_features)
        ws = [_f32c(w) for w in params]
        rn = [(_f32c(m), _f32c(v)) for m, v in running]
        L = _lib.lib()
        dev = cq.device
        group = _sync_group()
        with torch.cuda.device(dev):
            probe = _cfg(dims, rates, p_attn, p_aspp, seed, None)
            nsaved = L.axvs_cc_module_train_saved_bytes(C.byref(probe))
            nscr = L.axvs_cc_module_train_scratch_bytes(C.byref(probe), 0)
            nstat = L.axvs_cc_module_train_bn_stats_floats(C.byref(probe))
            if nsaved == 0 or nscr == 0:
                raise RuntimeError("axvs_cc_module_train_saved_bytes: " + L.axvs_last_error().decode())
            saved = torch.empty(nsaved, dtype=torch.uint8, device=dev)
            scratch = torch.empty(nscr, dtype=torch.uint8, device=dev)
            logits = torch.empty(nl, 1, Q, K1, dtype=torch.float32, device=dev)
            masks = torch.empty(nl, B, Q, Tc * V, H, W, dtype=torch.float32, device=dev)
            stats = torch.empty(nstat, dtype=torch.float32, device=dev)
            hook = _AllReduce(scratch, group) if group is not None else None
            c = _cfg(dims, rates, p_attn, p_aspp, seed, hook)
            ptrs = [w.data_ptr() for w in ws]
            layers = (_lib.AxvsCCLayerParams * nl)(*[_layer_struct(ptrs[i * _PER_LAYER:(i + 1) * _PER_LAYER]) for i in range(nl)])
            heads = _head_struct(ptrs[nl * _PER_LAYER:], [(m.data_ptr(), v.data_ptr()) for m, v in rn])
            rc = L.axvs_cc_module_train_fwd(cq.data_ptr(), pf.data_ptr(), logits.data_ptr(), masks.data_ptr(), stats.data_ptr(), layers,
                                            C.byref(heads), C.byref(c), saved.data_ptr(), nsaved, scratch.data_ptr(), nscr, _stream(dev))
            if hook is not None and hook.error is not None:
                raise hook.error
            _lib.check(rc, "axvs_cc_module_train_fwd")
        ctx.save_for_backward(cq, pf, *ws)
        ctx.amp = _lib.current_amp()
        ctx.running = rn
        ctx.cfg = cfg
        ctx.saved_buf = saved
        ctx.in_dtypes = (clip_query.dtype, [w.dtype for w in params])
        ctx.mark_non_differentiable(stats)
        return logits, masks, stats

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_logits, d_masks, _d_stats):
        from .modules import _stream
        cq, pf, *ws = ctx.saved_tensors
        dims, rates, p_attn, p_aspp, seed = ctx.cfg
        B, Q, Tc, V, H, W, K1, nl = dims
        L = _lib.lib()
        dev = cq.device
        group = _sync_group()
        with torch.cuda.device(dev):
            gl = _f32c(d_logits) if d_logits is not None else torch.zeros(nl, 1, Q, K1, dtype=torch.float32, device=dev)
            gm = _f32c(d_masks) if d_masks is not None else torch.zeros(nl, B, Q, Tc * V, H, W, dtype=torch.float32, device=dev)
            probe = _cfg(dims, rates, p_attn, p_aspp, seed, None)
            nsaved = L.axvs_cc_module_train_saved_bytes(C.byref(probe))
            nscr = L.axvs_cc_module_train_scratch_bytes(C.byref(probe), 1)
            scratch = torch.empty(nscr, dtype=torch.uint8, device=dev)
            hook = _AllReduce(scratch, group) if group is not None else None
            c = _cfg(dims, rates, p_attn, p_aspp, seed, hook)
            sizes = [w.numel() for w in ws]
            flat = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
            grads, off = [], 0
            for w, n in zip(ws, sizes):
                grads.append(flat[off:off + n].view(w.shape))
                off += n
            d_cq = torch.empty_like(cq)
            ptrs = [w.data_ptr() for w in ws]
            gptrs = [g.data_ptr() for g in grads]
            layers = (_lib.AxvsCCLayerParams * nl)(*[_layer_struct(ptrs[i * _PER_LAYER:(i + 1) * _PER_LAYER]) for i in range(nl)])
            lgrads = (_lib.AxvsCCLayerParams * nl)(*[_layer_struct(gptrs[i * _PER_LAYER:(i + 1) * _PER_LAYER]) for i in range(nl)])
            heads = _head_struct(ptrs[nl * _PER_LAYER:], [(m.data_ptr(), v.data_ptr()) for m, v in ctx.running])
            hgrads = _head_grads(gptrs[nl * _PER_LAYER:])
            with _lib.train_amp(ctx.amp):
                rc = L.axvs_cc_module_train_bwd(gl.data_ptr(), gm.data_ptr(), cq.data_ptr(), pf.data_ptr(), layers, C.byref(heads), lgrads,
                                                C.byref(hgrads), d_cq.data_ptr(), C.byref(c), ctx.saved_buf.data_ptr(), nsaved, scratch.data_ptr(), nscr,
                                                _stream(dev))
            if hook is not None and hook.error is not None:
                raise hook.error
            _lib.check(rc, "axvs_cc_module_train_bwd")
        qd, wd = ctx.in_dtypes
        return (d_cq.to(qd), None, None, None, *[g.to(dt) for g, dt in zip(grads, wd)])


def cc_module_train(mod, clip_query: Tensor, panoptic_features: Tensor):
    """Differentiable forward of a CrossClipTrackingModule in train() mode -> (class logits [nl,1,Q,K1], mask logits [nl,B,Q,Tc*V,H,W]);
    updates the running statistics of the module's four BatchNorm sites like the reference's forward does (once per layer)."""
    B, Q, Tc, Cq = clip_query.shape
    V = mod.num_clip_frames
    if Cq != 256 or panoptic_features.dim() != 5 or panoptic_features.shape[1] != 128:
        raise RuntimeError("clip_query must be [B,Q,T,256] and panoptic_features [B,128,T*V,H,W]")
    Bp, _, TV, H, W = panoptic_features.shape
    if Bp != B or TV != Tc * V:
        raise RuntimeError(f"panoptic_features {tuple(panoptic_features.shape)} does not match clip_query {tuple(clip_query.shape)} / V={V}")
    if panoptic_features.requires_grad:
        raise NotImplementedError("axial_vs_amd: no gradient for panoptic_features (the frozen segmenter's output, maxtron_cc_model.py:104-108); "
                                  "detach() it")
    for lay in mod.transformer_trajectory_self_attention_layers:
        if lay.normalize_before or lay.dropout.p != 0.0:
            raise NotImplementedError("axial_vs_amd: the cross-clip layer is post-norm with dropout 0 (CC:249-255)")
    bns = _bn_modules(mod)
    K1 = mod._predictor._transformer_class_head.conv.weight.shape[0]
    nl = mod.num_layers
    p_attn, p_aspp = float(mod.attn_drop), float(mod.aspp_drop)
    seed = getattr(mod, "dropout_seed", None)
    if seed is None:
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if (p_attn > 0 or p_aspp > 0) else 0
    cfg = ((int(B), int(Q), int(Tc), int(V), int(H), int(W), int(K1), int(nl)), tuple(int(r) for r in mod.atrous_rates), p_attn, p_aspp, int(seed))
    running = [(bn.running_mean, bn.running_var) for bn in bns]
    args = (clip_query, panoptic_features, cfg, running, *module_parameters(mod))
    if torch.is_autocast_enabled():
        amp = _lib.autocast_mode(mod)       # (read before autocast is switched off for the call)
        with torch.autocast(device_type="cuda", enabled=False), _lib.train_amp(amp):
            logits, masks, stats = _CCModuleTrain.apply(*args)
    else:
        logits, masks, stats = _CCModuleTrain.apply(*args)
    # running statistics: one momentum step per layer call, in layer order (nn.BatchNorm semantics, momentum 0.01), folded into
    # one update per buffer: r <- (1-m)^nl r + sum_l m (1-m)^(nl-1-l) stat_l
    with torch.no_grad():
        off = 0
        for bn in bns:
            Cn = bn.num_features
            st = stats[off:off + nl * 2 * Cn].view(nl, 2, Cn)
            off += nl * 2 * Cn
            if not bn.track_running_stats or bn.running_mean is None:
                continue
            if bn.momentum is None:          # cumulative moving average: the factor depends on the counter, step by step
                for l in range(nl):
                    bn.num_batches_tracked += 1
                    m = 1.0 / float(bn.num_batches_tracked)
                    bn.running_mean.mul_(1 - m).add_(st[l, 0].to(bn.running_mean.dtype), alpha=m)
                    bn.running_var.mul_(1 - m).add_(st[l, 1].to(bn.running_var.dtype), alpha=m)
                continue
            m = float(bn.momentum)
            coef = torch.tensor([m * (1 - m) ** (nl - 1 - l) for l in range(nl)], dtype=torch.float32, device=st.device)
            upd = torch.einsum("l,lsc->sc", coef, st)
            bn.running_mean.mul_((1 - m) ** nl).add_(upd[0].to(bn.running_mean.dtype))
            bn.running_var.mul_((1 - m) ** nl).add_(upd[1].to(bn.running_var.dtype))
            bn.num_batches_tracked += nl
    return logits, masks


# ---- the layer chain alone: the Tube-Link head's train() mode (its prediction heads are torch modules around it) ----------------------
def chain_parameters(mod, num_layers: int) -> List[Tensor]:
    ps: List[Tensor] = []
    for i in range(num_layers):
        lay, asp, cn = mod.transformer_trajectory_self_attention_layers[i], mod.conv_short_aggregate_layers[i], mod.conv_norms[i]
        at = lay.self_attn
        ps += [at.qkv.weight, at.qkv.bias, at.proj_q.weight, at.proj_q.bias, at.proj_kv.weight, at.proj_kv.bias, at.proj.weight, at.proj.bias,
               lay.norm.weight, lay.norm.bias]
        for k in range(3):
            conv = getattr(asp, f"_aspp_conv{k}")
            ps += [conv.weight, conv.bias]
        ps += [asp._proj_conv_bn_act.conv.weight, asp._proj_conv_bn_act.norm.weight, asp._proj_conv_bn_act.norm.bias, cn.weight, cn.bias]
    return ps


class _CCLayersTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, clip_query, cfg, *params):
        from .modules import _stream
        dims, rates, p_attn, p_aspp, seed = cfg
        if not clip_query.is_cuda:
            raise RuntimeError("axial_vs_amd: the training tier needs GPU tensors; there is no CPU fallback")
        B, Q, Tc, nl = dims
        cq = _f32c(clip_query)
        ws = [_f32c(w) for w in params]
        L = _lib.lib()
        dev = cq.device
        full = (B, Q, Tc, 1, 1, 1, 1, nl)
        with torch.cuda.device(dev):
            c = _cfg(full, rates, p_attn, p_aspp, seed, None)
            c.chain_only = 1
            nsaved = L.axvs_cc_module_train_saved_bytes(C.byref(c))
            nscr = L.axvs_cc_module_train_scratch_bytes(C.byref(c), 0)
            if nsaved == 0 or nscr == 0:
                raise RuntimeError("axvs_cc_module_train_saved_bytes: " + L.axvs_last_error().decode())
            saved = torch.empty(nsaved, dtype=torch.uint8, device=dev)
            scratch = torch.empty(nscr, dtype=torch.uint8, device=dev)
            out = torch.empty(nl, B, Q, Tc, 256, dtype=torch.float32, device=dev)
            ptrs = [w.data_ptr() for w in ws]
            layers = (_lib.AxvsCCLayerParams * nl)(*[_layer_struct(ptrs[i * _PER_LAYER:(i + 1) * _PER_LAYER]) for i in range(nl)])
            _lib.check(L.axvs_cc_layers_train_fwd(cq.data_ptr(), out.data_ptr(), layers, C.byref(c), saved.data_ptr(), nsaved, scratch.data_ptr(), nscr,
                                                  _stream(dev)), "axvs_cc_layers_train_fwd")
        ctx.save_for_backward(cq, *ws)
        ctx.amp = _lib.current_amp()
        ctx.cfg = (full, rates, p_attn, p_aspp, seed)
        ctx.saved_buf = saved
        ctx.in_dtypes = (clip_query.dtype, [w.dtype for w in params])
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_out):
        from .modules import _stream
        cq, *ws = ctx.saved_tensors
        full, rates, p_attn, p_aspp, seed = ctx.cfg
        nl = full[-1]
        L = _lib.lib()
        dev = cq.device
        with torch.cuda.device(dev):
            g = _f32c(d_out)
            c = _cfg(full, rates, p_attn, p_aspp, seed, None)
            c.chain_only = 1
            nsaved = L.axvs_cc_module_train_saved_bytes(C.byref(c))
            nscr = L.axvs_cc_module_train_scratch_bytes(C.byref(c), 1)
            scratch = torch.empty(nscr, dtype=torch.uint8, device=dev)
            sizes = [w.numel() for w in ws]
            flat = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
            grads, off = [], 0
            for w, n in zip(ws, sizes):
                grads.append(flat[off:off + n].view(w.shape))
                off += n
            d_cq = torch.empty_like(cq)
            ptrs, gptrs = [w.data_ptr() for w in ws], [t.data_ptr() for t in grads]
            layers = (_lib.AxvsCCLayerParams * nl)(*[_layer_struct(ptrs[i * _PER_LAYER:(i + 1) * _PER_LAYER]) for i in range(nl)])
            lgrads = (_lib.AxvsCCLayerParams * nl)(*[_layer_struct(gptrs[i * _PER_LAYER:(i + 1) * _PER_LAYER]) for i in range(nl)])
            with _lib.train_amp(ctx.amp):
                _lib.check(L.axvs_cc_layers_train_bwd(g.data_ptr(), cq.data_ptr(), layers, lgrads, d_cq.data_ptr(), C.byref(c), ctx.saved_buf.data_ptr(),
                                                      nsaved, scratch.data_ptr(), nscr, _stream(dev)), "axvs_cc_layers_train_bwd")
        qd, wd = ctx.in_dtypes
        return (d_cq.to(qd), None, *[t.to(dt) for t, dt in zip(grads, wd)])


def cc_layers_train(mod, clip_query: Tensor, num_layers: int, rates, p_attn: float, p_aspp: float) -> Tensor:
    """Differentiable layer chain of a cross-clip module (clip_query [B,Q,Tc,256]) -> the clip queries after every layer [nl,B,Q,Tc,256]."""
    B, Q, Tc, Cq = clip_query.shape
    if Cq != 256:
        raise RuntimeError("clip_query must be [B,Q,Tc,256]")
    seed = getattr(mod, "dropout_seed", None)
    if seed is None:
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if (p_attn > 0 or p_aspp > 0) else 0
    cfg = ((int(B), int(Q), int(Tc), int(num_layers)), tuple(int(r) for r in rates), float(p_attn), float(p_aspp), int(seed))
    args = (clip_query, cfg, *chain_parameters(mod, num_layers))
    if torch.is_autocast_enabled():
        amp = _lib.autocast_mode(mod)       # (read before autocast is switched off for the call)
        with torch.autocast(device_type="cuda", enabled=False), _lib.train_amp(amp):
            return _CCLayersTrain.apply(*args)
    return _CCLayersTrain.apply(*args)
