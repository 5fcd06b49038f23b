"""nn.Module mirror of the cross-clip tracking module.

Reference: CC = MaXTron_Video-kMaX/maxtron_deeplab/modeling/cross_clip_tracking_module/maxtron_cross_clip_tracking_module.py
(same classes inlined in MaXTron_Tube-Link/models/video/tube_link_vis/mask2former_video_cc_head.py:125-247).
Same constructor, attribute names and state-dict keys as `CrossClipTrackingModule` (CC:204-331) and its parts
(`TrajectoryAttention` CC:78-130, `TrajectoryAttentionLayer` :133-173, `ASPP` :176-201, `MaXTronCCPredictor` :30-75,
`ConvBN` / channels-first `LayerNorm` from kmax_deeplab); forward runs in libaxvs.so.  norm_fn='ln' (every shipped config),
kernel sizes 3.  eval(): the fused 16-bit inference kernels; train(): the fp32 training tier with a backward pass (cc_training.py).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor

from . import _lib
from .modules import _dev_f32, _param_key, _require_eval, _select_sync_words, _stream, _traj_struct, _workspace, _guarded


class _LayerNormCF(nn.Module):
    """channels-first LayerNorm parameters (kmax_deeplab/modeling/backbone/convnext.py:52-81), eps 1e-6."""

    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.eps = 1e-6


class ConvBN(nn.Module):
    """Parameter container with the reference's names: .conv (Conv1d), .norm (BatchNorm / channels-first LN / Identity)."""

    def __init__(self, cin, cout, kernel_size=1, bias=True, norm=None, act=None):
        super().__init__()
        self.conv = nn.Conv1d(cin, cout, kernel_size=kernel_size, bias=bias)
        if norm is None or norm == "none":
            self.norm = nn.Identity()
        elif norm == "syncbn":
            self.norm = nn.BatchNorm1d(cout, eps=1e-3, momentum=0.01)   # same buffers/keys as nn.SyncBatchNorm
        elif norm == "ln":
            self.norm = _LayerNormCF(cout)
        else:
            raise NotImplementedError(norm)
        self.act = nn.GELU() if act == "gelu" else nn.Identity()


class TrajectoryAttention(nn.Module):       # CC:78-89
    def __init__(self, d_model, nhead, attn_drop):
        super().__init__()
        self.num_heads = nhead
        self.head_dim = d_model // nhead
        self.scale = self.head_dim ** -0.5
        self.qkv = nn.Linear(d_model, d_model * 3)
        self.proj_q = nn.Linear(d_model, d_model)
        self.proj_kv = nn.Linear(d_model, d_model * 2)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(d_model, d_model)


class TrajectoryAttentionLayer(nn.Module):  # CC:133-153
    def __init__(self, d_model, nhead, dropout=0.0, attn_drop=0.0, activation="relu", normalize_before=False):
        super().__init__()
        self.self_attn = TrajectoryAttention(d_model, nhead, attn_drop=attn_drop)
        self.norm = nn.LayerNorm(d_model)
        self.dropout = nn.Dropout(dropout)
        self.normalize_before = normalize_before
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)


class ASPP(nn.Module):                       # CC:176-190
    def __init__(self, in_channels, output_channels, kernel_sizes, atrous_rates, dropout_rate, norm_fn):
        super().__init__()
        if list(kernel_sizes) != [3, 3, 3]:
            raise NotImplementedError("axial_vs_amd: ASPP kernel sizes other than [3,3,3] have no HIP path")
        # the reference's ConvBN can be built with 'ln' and 'syncbn' (kmax_pixel_decoder.py:32-40, :68-69: 'none' / None fail in its __init__, 'bn' is unknown to
        # get_norm); every shipped config uses 'ln'.  'syncbn' runs in eval mode only (running statistics folded into a per-channel scale / shift: library
        # option cc_aspp_affine); its train() mode has no HIP path
        if norm_fn not in ("ln", "syncbn"):
            raise NotImplementedError(f"axial_vs_amd: ASPP norm_fn {norm_fn!r} (the reference's ConvBN builds with 'ln' or 'syncbn')")
        self.norm_fn = norm_fn
        for i in range(3):
            setattr(self, f"_aspp_conv{i}", nn.Conv1d(in_channels, output_channels, kernel_size=3, dilation=atrous_rates[i],
                                                      padding="same", padding_mode="replicate"))
        self._proj_conv_bn_act = ConvBN(output_channels * 3, output_channels, 1, bias=False, norm=norm_fn, act="gelu")
        self._proj_drop = nn.Dropout(p=dropout_rate)


class MaXTronCCPredictor(nn.Module):         # CC:30-43
    def __init__(self, num_classes=133 + 1):
        super().__init__()
        self._transformer_mask_head = ConvBN(256, 128, 1, bias=False, norm="syncbn")
        self._transformer_class_head = ConvBN(256, num_classes, 1, norm=None)
        self._transformer_class_activation_head = ConvBN(256, 1, 1, norm=None)
        self._pixel_space_mask_batch_norm = nn.BatchNorm1d(1, eps=1e-3, momentum=0.01)
        nn.init.constant_(self._pixel_space_mask_batch_norm.weight, 0.1)


def _bn(m) -> _lib.AxvsBN:
    return m.weight, m.bias, m.running_mean, m.running_var


def _pack_cc_layers(mod, num_layers, f, keep, dt, dev):
    """axvs_cc_layer_pack for every (trajectory layer, ASPP, LayerNorm) triple of `mod` -> list of packed device buffers."""
    L = _lib.lib()
    layers = []
    for i in range(num_layers):
        lay, asp, cn = mod.transformer_trajectory_self_attention_layers[i], mod.conv_short_aggregate_layers[i], mod.conv_norms[i]
        ps = _lib.AxvsCCLayerParams()
        ps.attn = _traj_struct(lay.self_attn, keep)
        ps.norm_w, ps.norm_b = f(lay.norm.weight), f(lay.norm.bias)
        for k in range(3):
            conv = getattr(asp, f"_aspp_conv{k}")
            ps.aspp_w[k], ps.aspp_b[k] = f(conv.weight), f(conv.bias)
        ps.aspp_proj_w = f(asp._proj_conv_bn_act.conv.weight)
        nrm = asp._proj_conv_bn_act.norm
        if isinstance(nrm, nn.BatchNorm1d):      # 'syncbn', eval mode: y * scale + shift with the running statistics (eps 1e-3)
            scale = (nrm.weight.detach().float() * torch.rsqrt(nrm.running_var.float() + nrm.eps)).contiguous()
            shift = (nrm.bias.detach().float() - nrm.running_mean.float() * scale).contiguous()
            ps.aspp_norm_w, ps.aspp_norm_b = f(scale), f(shift)
        else:
            ps.aspp_norm_w, ps.aspp_norm_b = f(nrm.weight), f(nrm.bias)
        ps.conv_norm_w, ps.conv_norm_b = f(cn.weight), f(cn.bias)
        buf = torch.empty(L.axvs_cc_layer_packed_bytes(), dtype=torch.uint8, device=dev)
        _lib.check(L.axvs_cc_layer_pack(C.byref(ps), buf.data_ptr(), _lib.DTYPES[dt], _stream(dev)), "axvs_cc_layer_pack")
        layers.append(buf)
    return layers


class CrossClipTrackingModule(nn.Module):
    def __init__(self, *, num_layers: int, num_classes: int, attn_drop: float, aspp_drop: float, kernel_sizes: List[int],
                 atrous_rates: List[int], norm_fn: str, num_clip_frames: int, mfma_dtype: Optional[str] = None):
        super().__init__()
        self.kernel_sizes = kernel_sizes
        self.atrous_rates = atrous_rates
        self.attn_drop = attn_drop
        self.aspp_drop = aspp_drop
        self.norm_fn = norm_fn
        self.num_clip_frames = num_clip_frames
        self.num_heads = 8
        self.num_layers = num_layers
        self.transformer_trajectory_self_attention_layers = nn.ModuleList()
        self.conv_short_aggregate_layers = nn.ModuleList()
        self.conv_norms = nn.ModuleList()
        for _ in range(num_layers):
            self.transformer_trajectory_self_attention_layers.append(
                TrajectoryAttentionLayer(d_model=256, nhead=8, dropout=0.0, attn_drop=attn_drop, normalize_before=False))
            self.conv_short_aggregate_layers.append(ASPP(256, 256, kernel_sizes, atrous_rates, aspp_drop, norm_fn))
            self.conv_norms.append(nn.LayerNorm(256))
        self._class_embedding_projection = ConvBN(256, 256, 1, bias=False, norm="syncbn", act="gelu")
        self._mask_embedding_projection = ConvBN(256, 256, 1, bias=False, norm="syncbn", act="gelu")
        self._predictor = MaXTronCCPredictor(num_classes=num_classes + 1)
        self.mfma_dtype = mfma_dtype
        self.eval_outputs_on_cpu = True     # the reference's eval branch returns CPU tensors (CC:59,70)
        # eval mode: True (default) = every layer's predictions are computed and returned under 'aux_outputs', as the reference does
        # (CC:283-322); False = only the last layer's predictor heads run and 'aux_outputs' is [] -- what the reference's own inference
        # path keeps (maxtron_cc_model.py reads aux_outputs under self.training only): 3/4 of the mask einsum's HBM writes less
        self.eval_aux_outputs = True
        self._packed = None
        self._packed_key = None

    # ---- packing -------------------------------------------------------------------------------------------------
    def _dtype(self) -> str:
        from . import modules
        return self.mfma_dtype or modules.default_operand_dtype()

    def _pack(self):
        dt = self._dtype()
        key = _param_key(self, dt) + tuple((b.data_ptr(), b._version) for b in self.buffers())
        if self._packed is not None and key == self._packed_key:
            return self._packed
        L = _lib.lib()
        dev = self.conv_norms[0].weight.device
        keep: list = []

        def f(t):
            tt = _dev_f32(t.detach(), "parameter")
            keep.append(tt)
            return tt.data_ptr()

        layers = _pack_cc_layers(self, self.num_layers, f, keep, dt, dev)
        pr = self._predictor
        K1 = pr._transformer_class_head.conv.weight.shape[0]
        hp = _lib.AxvsCCHeadParams()
        hp.class_proj_w = f(self._class_embedding_projection.conv.weight)
        hp.class_proj_bn = _lib.AxvsBN(*[f(t) for t in _bn(self._class_embedding_projection.norm)])
        hp.mask_proj_w = f(self._mask_embedding_projection.conv.weight)
        hp.mask_proj_bn = _lib.AxvsBN(*[f(t) for t in _bn(self._mask_embedding_projection.norm)])
        hp.mask_head_w = f(pr._transformer_mask_head.conv.weight)
        hp.mask_head_bn = _lib.AxvsBN(*[f(t) for t in _bn(pr._transformer_mask_head.norm)])
        hp.class_head_w, hp.class_head_b = f(pr._transformer_class_head.conv.weight), f(pr._transformer_class_head.conv.bias)
        hp.act_head_w = f(pr._transformer_class_activation_head.conv.weight)
        hp.act_head_b = f(pr._transformer_class_activation_head.conv.bias)
        hp.pixel_bn = _lib.AxvsBN(*[f(t) for t in _bn(pr._pixel_space_mask_batch_norm)])
        hbuf = torch.empty(L.axvs_cc_heads_packed_bytes(K1), dtype=torch.uint8, device=dev)
        _lib.check(L.axvs_cc_heads_pack(C.byref(hp), hbuf.data_ptr(), K1, _lib.DTYPES[dt], _stream(dev)), "axvs_cc_heads_pack")
        torch.cuda.current_stream(dev).synchronize()      # `keep` (fp32 staging copies) may be released after this
        self._packed, self._packed_key = (layers, hbuf, K1), key
        return self._packed

    # ---- forward (CC:275-322) ------------------------------------------------------------------------------------
    def _forward_train(self, clip_query: Tensor, panoptic_features: Tensor):
        """train() mode (CC:53-57 instead of :58-70): differentiable, BatchNorm on batch statistics, outputs stay on the GPU."""
        from .cc_training import cc_module_train
        logits, masks = cc_module_train(self, clip_query, panoptic_features)
        # unbind: ONE autograd node that stacks the per-layer gradients once (indexing layer by layer would add a full-size zero
        # tensor per layer in backward: 4 x 3.2 GB at the shipped VIPSeg shape)
        cls_all = list(logits.unbind(0))
        mask_all = list(masks.unbind(0))
        # (_set_aux_loss resamples the auxiliary masks to the last layer's size, CC:324-331: every layer has that size already)
        aux = [{"pred_logits": a, "pred_masks": b} for a, b in zip(cls_all[:-1], mask_all[:-1])]
        return {"pred_logits": cls_all[-1], "pred_masks": mask_all[-1], "aux_outputs": aux}

    @_guarded
    def forward(self, clip_query: Tensor, panoptic_features: Tensor):
        if self.training:
            if self.norm_fn != "ln":
                raise NotImplementedError("axial_vs_amd: train() mode of the cross-clip module needs norm_fn='ln' (the shipped configuration)")
            return self._forward_train(clip_query, panoptic_features)
        cq = _dev_f32(clip_query, "clip_query")
        pf = _dev_f32(panoptic_features, "panoptic_features")
        B, Q, Tc, Cq = cq.shape
        if Cq != 256 or pf.shape[1] != 128:
            raise RuntimeError("clip_query must be [B,Q,T,256] and panoptic_features [B,128,T*V,H,W]")
        V = self.num_clip_frames
        Bp, _, TV, H, W = pf.shape
        if Bp != B or TV != Tc * V:
            raise RuntimeError(f"panoptic_features {tuple(pf.shape)} does not match clip_query {tuple(cq.shape)} / V={V}")
        L = _lib.lib()
        layers, hbuf, K1 = self._pack()
        dt = _lib.DTYPES[self._dtype()]
        dev = cq.device
        nl = self.num_layers
        rates = (C.c_int * 3)(*[int(r) for r in self.atrous_rates])
        # ONE library call runs the whole layer loop (CC:283-318): the layer chain first (layer i+1 only needs layer i's clip
        # queries, not its predictions), then the predictor heads of all layers at once (their weights are shared across layers) and
        # the mask einsum of all layers in one pass over the pixel features.
        main = torch.cuda.current_stream(dev)
        ws = _workspace(dev, L.axvs_cc_module_workspace_bytes(B, Q, Tc, nl))
        nh = nl if self.eval_aux_outputs else 1          # layers whose predictor heads run (the last `nh`)
        logits = torch.empty(nh, 1, Q, K1, dtype=torch.float32, device=dev)
        masks = torch.empty(nh, B, Q, TV, H, W, dtype=torch.float32, device=dev)
        last = torch.empty_like(cq)
        pl = (C.c_void_p * nl)(*[b.data_ptr() for b in layers])
        _select_sync_words(dev)          # the layers' trajectory attention runs q/k/v + attention as one launch (include/axvs.h)
        if nh != nl:
            _lib.check(L.axvs_set_option(b"cc_last_heads_only", 1), "axvs_set_option")
        affine = self.norm_fn != "ln"
        if affine:
            _lib.check(L.axvs_set_option(b"cc_aspp_affine", 1), "axvs_set_option")
        try:
            _lib.check(L.axvs_cc_module_fwd(cq.data_ptr(), pf.data_ptr(), logits.data_ptr(), masks.data_ptr(), last.data_ptr(), pl, hbuf.data_ptr(),
                                            nl, B, Q, Tc, V, H, W, K1, rates, dt, ws.data_ptr(), ws.numel(), main.cuda_stream),
                       "axvs_cc_module_fwd")
        finally:
            if nh != nl:
                L.axvs_set_option(b"cc_last_heads_only", 0)
            if affine:
                L.axvs_set_option(b"cc_aspp_affine", 0)
        cur = last
        cls_all = [logits[i] for i in range(nh)]
        mask_all = [masks[i] for i in range(nh)]
        if self.eval_outputs_on_cpu:
            cls_all = [c.cpu() for c in cls_all]
            mask_all = [m.cpu() for m in mask_all]
        size = mask_all[-1].shape[-3:]
        ac = size[-1] % 2 == 1
        aux = [{"pred_logits": a, "pred_masks": b if b.shape[-3:] == size else F.interpolate(b, size=size, mode="trilinear", align_corners=ac)}
               for a, b in zip(cls_all[:-1], mask_all[:-1])]
        self.last_clip_query = cur
        return {"pred_logits": cls_all[-1], "pred_masks": mask_all[-1], "aux_outputs": aux}


class TubeLinkCrossClipHead(nn.Module):
    """The cross-clip members of Tube-Link's `Mask2FormerVideoCCHeadTube` (TLCC = MaXTron_Tube-Link/models/video/tube_link_vis/
    mask2former_video_cc_head.py:365-394) under their own names -- a checkpoint of the head loads with strict=False and fills
    every parameter here -- and the cross-clip part of its forward: the layer loop (TLCC:925-946), `forward_head_clips`
    (:761-781) and `pred_class` (:783-797).  Keyword names follow the head's constructor (:283-330).

    forward(clip_query [B,Tc,Q,256] (the matched clip queries, TLCC:919), mask_features [B,T,Cm,h,w], T = Tc*frames_per_clip)
        -> (tuple of class logits [B,Q,K+1] per layer, tuple of mask logits [B,T,Q,h,w] per layer)
    """

    def __init__(self, *, num_classes: int, feat_channels: int = 256, out_channels: int = 256, num_cc_layers: int = 6,
                 trajectory_drop_out: float = 0.0, kernel_sizes=(3, 3, 3), atrous_rates=(1, 2, 3), drop_path_prob: float = 0.1,
                 aspp_norm_fn: Optional[str] = "ln", mfma_dtype: Optional[str] = None):
        super().__init__()
        if feat_channels != 256:
            raise NotImplementedError("axial_vs_amd: the cross-clip kernels are built for feat_channels = 256")
        if out_channels not in (128, 256):
            raise NotImplementedError("axial_vs_amd: mask feature channels must be 128 or 256")
        self.num_classes, self.num_cc_layers = num_classes, num_cc_layers
        self.kernel_sizes, self.atrous_rates = kernel_sizes, atrous_rates
        self.transformer_decoder = nn.Module()
        self.transformer_decoder.post_norm = nn.LayerNorm(feat_channels)
        self.cls_embed = nn.Linear(feat_channels, num_classes + 1)
        self.mask_embed = nn.Sequential(nn.Linear(feat_channels, feat_channels), nn.ReLU(inplace=True),
                                        nn.Linear(feat_channels, feat_channels), nn.ReLU(inplace=True),
                                        nn.Linear(feat_channels, out_channels))
        self.transformer_trajectory_self_attention_layers = nn.ModuleList()
        self.conv_short_aggregate_layers = nn.ModuleList()
        self.conv_norms = nn.ModuleList()
        for _ in range(num_cc_layers):
            self.transformer_trajectory_self_attention_layers.append(
                TrajectoryAttentionLayer(d_model=256, nhead=8, dropout=0.0, attn_drop=trajectory_drop_out, normalize_before=False))
            self.conv_short_aggregate_layers.append(ASPP(256, 256, list(kernel_sizes), list(atrous_rates), drop_path_prob, aspp_norm_fn))
            self.conv_norms.append(nn.LayerNorm(256))
        self.activation_proj = nn.Linear(256, 1)
        self.mfma_dtype = mfma_dtype
        self._packed = None
        self._packed_key = None

    def _dtype(self) -> str:
        from . import modules
        return self.mfma_dtype or modules.default_operand_dtype()

    def _pack(self):
        dt = self._dtype()
        key = _param_key(self, dt) + tuple((b.data_ptr(), b._version) for b in self.buffers())     # (running statistics of aspp_norm_fn = 'syncbn')
        if self._packed is not None and key == self._packed_key:
            return self._packed
        L = _lib.lib()
        dev = self.activation_proj.weight.device
        keep: list = []

        def f(t):
            tt = _dev_f32(t.detach(), "parameter")
            keep.append(tt)
            return tt.data_ptr()

        layers = _pack_cc_layers(self, self.num_cc_layers, f, keep, dt, dev)
        K1, Cm = self.cls_embed.weight.shape[0], self.mask_embed[4].weight.shape[0]
        hp = _lib.AxvsTLHeadParams()
        pn = self.transformer_decoder.post_norm
        hp.post_norm_w, hp.post_norm_b = f(pn.weight), f(pn.bias)
        hp.activation_proj_w, hp.activation_proj_b = f(self.activation_proj.weight), f(self.activation_proj.bias)
        hp.cls_embed_w, hp.cls_embed_b = f(self.cls_embed.weight), f(self.cls_embed.bias)
        for k, idx in enumerate((0, 2, 4)):
            hp.mask_embed_w[k], hp.mask_embed_b[k] = f(self.mask_embed[idx].weight), f(self.mask_embed[idx].bias)
        hbuf = torch.empty(L.axvs_tl_heads_packed_bytes(K1, Cm), dtype=torch.uint8, device=dev)
        _lib.check(L.axvs_tl_heads_pack(C.byref(hp), hbuf.data_ptr(), K1, Cm, _lib.DTYPES[dt], _stream(dev)), "axvs_tl_heads_pack")
        torch.cuda.current_stream(dev).synchronize()
        self._packed, self._packed_key = (layers, hbuf, K1, Cm), key
        return self._packed

    def _forward_train(self, clip_query: Tensor, mask_features: Tensor):
        """train() mode (TLCC:925-947 with forward_head_clips :761-781 and pred_class :783-797 under autograd): the layer chain on the
        library's training tier (axvs_cc_layers_train_fwd / _bwd), the prediction heads -- post_norm, class pooling, the mask MLP,
        the per-clip einsum -- as the reference's torch modules."""
        from .cc_training import cc_layers_train
        B, Tc, Q, _ = clip_query.shape
        T = mask_features.shape[1]
        fpc = T // Tc
        trj = self.transformer_trajectory_self_attention_layers[0].self_attn.attn_drop.p
        asp = self.conv_short_aggregate_layers[0]._proj_drop.p
        queries = cc_layers_train(self, clip_query.permute(0, 2, 1, 3).contiguous(), self.num_cc_layers, self.atrous_rates, trj, asp)
        cls_all, mask_all = [], []
        for i in range(self.num_cc_layers):
            xn = self.transformer_decoder.post_norm(queries[i]).permute(0, 2, 1, 3)              # [B,Tc,Q,C]   TLCC:768-769
            act = torch.softmax(self.activation_proj(xn), dim=1)                                  # softmax over the clips, :789-791
            cls_all.append(self.cls_embed((xn * act).sum(dim=1)))
            me = self.mask_embed(xn)                                                               # [B,Tc,Q,Cm]
            mask_all.append(torch.cat([torch.einsum("bqc,btchw->btqhw", me[:, c], mask_features[:, fpc * c:fpc * (c + 1)])
                                       for c in range(Tc)], dim=1))                                # :774-778
        return tuple(cls_all), tuple(mask_all)

    @_guarded
    def forward(self, clip_query: Tensor, mask_features: Tensor):
        if self.training:
            if self.conv_short_aggregate_layers[0].norm_fn != "ln":
                raise NotImplementedError("axial_vs_amd: train() mode of the cross-clip head needs aspp_norm_fn='ln' (the shipped configuration)")
            return self._forward_train(clip_query, mask_features)
        cq = _dev_f32(clip_query, "clip_query")
        mf = _dev_f32(mask_features, "mask_features")
        B, Tc, Q, Cq = cq.shape
        layers, hbuf, K1, Cm = self._pack()
        if Cq != 256 or mf.dim() != 5 or mf.shape[0] != B or mf.shape[2] != Cm or mf.shape[1] % Tc:
            raise RuntimeError(f"clip_query must be [B,Tc,Q,256] and mask_features [B,Tc*f,{Cm},h,w]; got {tuple(cq.shape)}, {tuple(mf.shape)}")
        T, h, w = mf.shape[1], mf.shape[3], mf.shape[4]
        L = _lib.lib()
        dt = _lib.DTYPES[self._dtype()]
        dev = cq.device
        nl = self.num_cc_layers
        rates = (C.c_int * 3)(*[int(r) for r in self.atrous_rates])
        main = torch.cuda.current_stream(dev)          # one library call for the whole loop (see CrossClipTrackingModule)
        ws = _workspace(dev, L.axvs_tl_cc_module_workspace_bytes(B, Q, Tc, Cm, nl))
        cur = cq.permute(0, 2, 1, 3).contiguous()          # [B,Q,Tc,C]: the token order of 'b c t q -> b (t q) c' (TLCC:931)
        logits = torch.empty(nl, B, Q, K1, dtype=torch.float32, device=dev)
        masks = torch.empty(nl, B, T, Q, h, w, dtype=torch.float32, device=dev)
        last = torch.empty_like(cur)
        pl = (C.c_void_p * nl)(*[b.data_ptr() for b in layers])
        _select_sync_words(dev)
        affine = self.conv_short_aggregate_layers[0].norm_fn != "ln"       # 'syncbn' (eval: folded running statistics)
        if affine:
            _lib.check(L.axvs_set_option(b"cc_aspp_affine", 1), "axvs_set_option")
        try:
            _lib.check(L.axvs_tl_cc_module_fwd(cur.data_ptr(), mf.data_ptr(), logits.data_ptr(), masks.data_ptr(), last.data_ptr(), pl, hbuf.data_ptr(),
                                               nl, B, Q, Tc, T // Tc, h, w, K1, Cm, rates, dt, ws.data_ptr(), ws.numel(), main.cuda_stream),
                       "axvs_tl_cc_module_fwd")
        finally:
            if affine:
                L.axvs_set_option(b"cc_aspp_affine", 0)
        cls_all = [logits[i] for i in range(nl)]
        mask_all = [masks[i] for i in range(nl)]
        return tuple(cls_all), tuple(mask_all)
