"""nn.Module mirrors of the within-clip pixel decoder around the stages (SURVEY 8f-2, 8b registry hook).

Reference: WC = MaXTron_Video-kMaX/maxtron_deeplab/modeling/within_clip_tracking_module
  `PositionEmbeddingSine` (WC/pos_embeddings.py:12-53), `MSDeformAttnTransformerEncoder` (WC/msdeformattn.py:219-275),
  `MSDeformAttnTransformerEncoderOnly` (:34-174), `MSDeformAttnPixelDecoder` (:293-437),
  `WithinClipTrackingModule` (WC/maxtron_within_clip_tracking_module.py:14-69).
Same constructor keywords, attribute names and state-dict keys (a reference checkpoint loads with strict=True); every
tensor op of `forward_features` runs in libaxvs.so (1x1 conv + GroupNorm projections straight from / to the backbone's NCHW
maps, sine embeddings, deformable spatial layers, axial-trajectory temporal layers); PyTorch only slices and concatenates
token buffers.  Eval only; configurations with spatial layers (with or without temporal layers), like every shipped config.
"""
from __future__ import annotations

import copy
import ctypes as C
import math
from typing import Dict, List, Optional

import torch
import torch.nn as nn
from torch import Tensor

from . import _lib
from .modules import PositionEmbeddingSine3D, TemporalEncoder, _dev_f32, _has_hooks, _param_key, _require_eval, _stream, _workspace, _guarded
from .msda import MSDeformAttn, MSDeformAttnTransformerEncoderLayer


def _get_clones(module, N):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(N)])


class PositionEmbeddingSine(nn.Module):
    """WC/pos_embeddings.py:12-53 (mask=None): returns [N, 2*num_pos_feats, H, W] like the reference."""

    def __init__(self, num_pos_feats=64, temperature=10000, normalize=False, scale=None):
        super().__init__()
        if scale is not None and normalize is False:
            raise ValueError("normalize should be True if scale is passed")
        self.num_pos_feats, self.temperature, self.normalize = num_pos_feats, temperature, normalize
        self.scale = 2 * math.pi if scale is None else scale

    def tokens_into(self, pos: Tensor, add: Optional[Tensor], N: int, H: int, W: int, row0: int) -> None:
        """pos[n, row0 + y*W + x, :] (+ add) for a level inside a [N, S, C] fp32 CUDA buffer."""
        Cc = 2 * self.num_pos_feats
        _lib.check(_lib.lib().axvs_pos2d(pos.data_ptr(), add.data_ptr() if add is not None else None, N, H, W, Cc, pos.shape[1], row0,
                                         float(self.temperature), int(self.normalize), float(self.scale), _stream(pos.device)), "axvs_pos2d")

    def forward(self, x, mask=None):
        if mask is not None:
            raise NotImplementedError("axial_vs_amd: PositionEmbeddingSine supports mask=None (the within-clip module's only use)")
        if not x.is_cuda:
            raise RuntimeError("axial_vs_amd: CUDA tensors only (no CPU fallback)")
        N, _, H, W = x.shape
        pos = torch.empty(N, H * W, 2 * self.num_pos_feats, dtype=torch.float32, device=x.device)
        self.tokens_into(pos, None, N, H, W, 0)
        return pos.view(N, H, W, -1).permute(0, 3, 1, 2)


def _conv_gn(in_channels: int, out_channels: int) -> nn.Sequential:
    """Parameter holder with the reference's names (`.0` conv, `.1` GroupNorm(32)); the computation is axvs_conv1x1_gn_fwd."""
    return nn.Sequential(nn.Conv2d(in_channels, out_channels, kernel_size=1), nn.GroupNorm(32, out_channels))


_side_streams: Dict[str, list] = {}
_IN_PLACE_LEVELS = True      # eval: the temporal levels of a stage are processed in place in the token buffer (tests switch it off to compare)


def _run_levels_concurrently(fns):
    """Independent small chains side by side: every fn but the last runs on a side stream of its own beside the last one on the caller's stream --
    fork / join with events, no host synchronisation; each stream has its own workspace (modules._workspace is keyed by stream).  Used for
    the temporal encoder of a stage, applied to each temporal level on its own (WC/msdeformattn.py:258-264: a Python loop over independent levels
    of 16 and 64 row tiles at BASELINE config 3, on a 256-CU chip), and (round 6) for the 1x1 convolution + GroupNorm projections of the levels
    (WC/msdeformattn.py:404-435: three independent GEMM -> statistics -> normalise chains of 30 - 40 us each).  Returns the results in order."""
    if len(fns) < 2:
        return [f() for f in fns]
    cur = torch.cuda.current_stream()
    key = str(cur.device)
    sides = _side_streams.setdefault(key, [])
    while len(sides) < len(fns) - 1:
        sides.append(torch.cuda.Stream(cur.device))
    res = []
    try:                     # (torch.cuda.set_stream, not the `with torch.cuda.stream` context: ~8 us less host time per switch, and config 3's
        for f, side in zip(fns[:-1], sides):     # forward is bound by its ~0.85 ms of host enqueue time on the slower hosts of the pool)
            side.wait_stream(cur)
            torch.cuda.set_stream(side)
            res.append(f())
    finally:
        torch.cuda.set_stream(cur)
    res.append(fns[-1]())
    for r, side in zip(res[:-1], sides):
        cur.wait_stream(side)
        for t in (r if isinstance(r, (tuple, list)) else (r,)):
            if isinstance(t, torch.Tensor):
                t.record_stream(cur)
    return res


class MSDeformAttnTransformerEncoder(nn.Module):
    def __init__(self, spatial_layer, num_stages, transformer_num_spatial_feature_levels, transformer_num_temporal_feature_levels=0,
                 temporal_layer=None):
        super().__init__()
        self.spatial_layers = _get_clones(spatial_layer, num_stages)
        self.transformer_num_spatial_feature_levels = transformer_num_spatial_feature_levels
        self.transformer_num_temporal_feature_levels = transformer_num_temporal_feature_levels
        if transformer_num_temporal_feature_levels > 0:
            self.temporal_layers = _get_clones(temporal_layer, num_stages)

    @staticmethod
    def get_reference_points(spatial_shapes, valid_ratios, device):
        """WC/msdeformattn.py:229-242 with all-valid masks (valid ratios = 1): pixel centres, repeated for every level."""
        refs = []
        for (H_, W_) in spatial_shapes:
            ys = (torch.arange(H_, dtype=torch.float32, device=device) + 0.5) / H_
            xs = (torch.arange(W_, dtype=torch.float32, device=device) + 0.5) / W_
            refs.append(torch.stack((xs.view(1, W_).expand(H_, W_).reshape(-1), ys.view(H_, 1).expand(H_, W_).reshape(-1)), -1))
        ref = torch.cat(refs, 0)
        return ref[None, :, None, :].expand(valid_ratios, -1, len(spatial_shapes), 2).contiguous()

    @_guarded
    def forward(self, src, spatial_shapes, level_start_index, valid_ratios, pos, padding_mask, pos_3d=None):
        """src / pos [BT, S, C]; spatial_shapes: list of (H, W); valid_ratios: BT (all maps valid); pos_3d: list of [B,T,H,W,C]."""
        if self.training or (torch.is_grad_enabled() and src.requires_grad):
            return self._forward_train(src, spatial_shapes, level_start_index, pos, padding_mask, pos_3d)
        output = src
        key = (tuple(spatial_shapes), src.shape[0], src.device)
        if getattr(self, "_ref_key", None) != key:
            self._ref, self._ref_key = self.get_reference_points(spatial_shapes, src.shape[0], src.device), key
        sizes = [h * w for h, w in spatial_shapes]
        h_attn = w_attn = None
        for i, spatial_layer in enumerate(self.spatial_layers):
            output = spatial_layer(output, pos, self._ref, spatial_shapes, level_start_index, padding_mask)
            if self.transformer_num_temporal_feature_levels > 0:
                # the temporal levels (the coarsest, first in the buffer) are replaced in place; the reference's split / cat
                # (WC/msdeformattn.py:258-264) would also copy the large level that passes through
                nt = self.transformer_num_temporal_feature_levels
                starts = [sum(sizes[:j]) for j in range(nt)]
                layer = self.temporal_layers[i]
                # weights are packed on THIS stream before the fork (both streams' launches read the same packed buffer)
                if hasattr(layer, "prepack"):
                    layer.prepack()
                # (a forward hook on the spatial layer or on this encoder may have captured `output`: the in-place levels would
                #  mutate the tensor it holds, so hooked modules keep the reference's split / cat data flow)
                if _IN_PLACE_LEVELS and output.is_contiguous() and hasattr(layer, "can_run_in_place") \
                        and not _has_hooks(spatial_layer) and not _has_hooks(self) \
                        and all(layer.can_run_in_place(pos_3d[j], frame_stride_rows=output.shape[1]) for j in range(nt)):
                    # the levels are read and written where they lie in the token buffer (frames S rows apart): no copies at all
                    _run_levels_concurrently([(lambda j=j: (layer.forward_level_in_place(output, starts[j], pos_3d[j]),))
                                              for j in range(nt)])
                    continue
                ins = [output[:, starts[j]:starts[j] + sizes[j]].contiguous() for j in range(nt)]
                res = _run_levels_concurrently([(lambda j=j: layer(src=ins[j], pos=pos_3d[j])) for j in range(nt)])
                for j in range(nt):
                    lvl, h_attn, w_attn = res[j]
                    output[:, starts[j]:starts[j] + sizes[j]] = lvl
        return output, h_attn, w_attn


    def _forward_train(self, src, spatial_shapes, level_start_index, pos, padding_mask, pos_3d):
        """train() mode: the reference's stage loop as it stands (WC/msdeformattn.py:244-273: split, temporal layers, cat -- out of
        place, one stream) around layers that run their own training tiers."""
        output = src
        key = (tuple(spatial_shapes), src.shape[0], src.device)
        if getattr(self, "_ref_key", None) != key:
            self._ref, self._ref_key = self.get_reference_points(spatial_shapes, src.shape[0], src.device), key
        sizes = [h * w for h, w in spatial_shapes]
        h_attn = w_attn = None
        for i, spatial_layer in enumerate(self.spatial_layers):
            output = spatial_layer(output, pos, self._ref, spatial_shapes, level_start_index, padding_mask)
            if self.transformer_num_temporal_feature_levels > 0:
                parts = list(torch.split(output, sizes, dim=1))
                for j in range(self.transformer_num_temporal_feature_levels):
                    parts[j], h_attn, w_attn = self.temporal_layers[i](src=parts[j].contiguous(), pos=pos_3d[j])
                output = torch.cat(parts, dim=1)
        return output, h_attn, w_attn


class TemporalTransformerEncoder(nn.Module):
    """Temporal-only stage loop (WC/msdeformattn.py:276-290): per stage, the temporal encoder on the coarsest levels; the other
    levels pass through.  Works on the concatenated [BT, S, C] token buffer like the spatial + temporal encoder."""

    def __init__(self, temporal_layer, num_stages, transformer_num_temporal_feature_levels=2):
        super().__init__()
        self.temporal_layers = _get_clones(temporal_layer, num_stages)
        self.transformer_num_temporal_feature_levels = transformer_num_temporal_feature_levels

    @_guarded
    def forward(self, src, spatial_shapes, pos_3d):
        sizes = [h * w for h, w in spatial_shapes]
        nt = self.transformer_num_temporal_feature_levels
        if self.training or (torch.is_grad_enabled() and src.requires_grad):      # the reference's loop (WC/msdeformattn.py:282-290), out of place
            parts = list(torch.split(src, sizes, dim=1))
            h_attn = w_attn = None
            for temporal_layer in self.temporal_layers:
                for j in range(nt):
                    parts[j], h_attn, w_attn = temporal_layer(src=parts[j].contiguous(), pos=pos_3d[j])
            return torch.cat(parts, dim=1), h_attn, w_attn
        if _IN_PLACE_LEVELS and all(hasattr(tl, "can_run_in_place") and tl.can_run_in_place(pos_3d[j], frame_stride_rows=src.shape[1])
                                    for tl in self.temporal_layers for j in range(nt)):
            out = src.contiguous().clone()              # (the caller's buffer is left alone); the levels are processed where they lie
            starts = [sum(sizes[:j]) for j in range(nt)]
            for temporal_layer in self.temporal_layers:
                temporal_layer.prepack()
                _run_levels_concurrently([(lambda j=j: (temporal_layer.forward_level_in_place(out, starts[j], pos_3d[j]),))
                                          for j in range(nt)])
            return out, None, None
        parts = [p.contiguous() for p in torch.split(src, sizes, dim=1)[:nt]]
        h_attn = w_attn = None
        for temporal_layer in self.temporal_layers:
            # weights are packed on the CALLER's stream before the fork: the side stream's level and this stream's level share the
            # packed buffer, and nothing else would order one stream's q/k/v launches behind the other stream's pack kernels
            if hasattr(temporal_layer, "prepack"):
                temporal_layer.prepack()
            res = _run_levels_concurrently([(lambda j=j: temporal_layer(src=parts[j], pos=pos_3d[j])) for j in range(nt)])
            for j in range(nt):
                parts[j], h_attn, w_attn = res[j]
        out = src.clone()                               # (the caller's buffer is left alone)
        start = 0
        for j in range(nt):
            out[:, start:start + sizes[j]] = parts[j]
            start += sizes[j]
        return out, h_attn, w_attn


class MSDeformAttnTransformerEncoderOnly(nn.Module):
    def __init__(self, d_model=256, nhead=8, num_stages=2, num_spatial_layers=2, num_temporal_layers=4,
                 temporal_attn_type="axial_trajectory", dim_feedforward=1024, dropout=0.1, attn_drop=0.1, activation="relu",
                 enc_n_points=4, num_spatial_feature_levels=3, num_temporal_feature_levels=2):
        super().__init__()
        self.d_model, self.nhead = d_model, nhead
        self.num_spatial_layers, self.num_temporal_layers = num_spatial_layers, num_temporal_layers
        assert num_spatial_layers + num_temporal_layers > 0, "number of layers should be greater than 0"
        if num_spatial_layers > 0:
            assert num_spatial_layers == num_stages, "number of spatial layers should be equal to number of stages"
            spatial_layer = MSDeformAttnTransformerEncoderLayer(d_model, dim_feedforward, dropout, activation, num_spatial_feature_levels,
                                                                nhead, enc_n_points)
        if num_temporal_layers > 0:
            temporal_layer = TemporalEncoder(d_model, dim_feedforward, dropout, attn_drop, activation, nhead, temporal_attn_type,
                                             num_temporal_layers // num_stages)
        if num_spatial_layers > 0 and num_temporal_layers > 0:
            self.encoder = MSDeformAttnTransformerEncoder(spatial_layer, num_spatial_layers, num_spatial_feature_levels,
                                                          num_temporal_feature_levels, temporal_layer)
        elif num_spatial_layers > 0:
            self.encoder = MSDeformAttnTransformerEncoder(spatial_layer, num_spatial_layers, num_spatial_feature_levels)
        else:                                                                   # temporal-only decoder (WC/msdeformattn.py:59-61)
            self.encoder = TemporalTransformerEncoder(temporal_layer, num_stages, num_temporal_feature_levels)
        if num_spatial_layers > 0:
            self.level_embed_2d = nn.Parameter(torch.Tensor(num_spatial_feature_levels, d_model))
        if num_temporal_layers > 0:
            self.level_embed_3d = nn.Parameter(torch.Tensor(num_temporal_feature_levels, d_model))
        self._reset_parameters()

    def _reset_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, MSDeformAttn):
                m._reset_parameters()
        if self.num_spatial_layers > 0:
            nn.init.normal_(self.level_embed_2d)
        if self.num_temporal_layers > 0:
            nn.init.normal_(self.level_embed_3d)


class MSDeformAttnPixelDecoder(nn.Module):
    def __init__(self, input_shape: Dict[str, object], *, transformer_dropout: float, transformer_attn_drop: float,
                 transformer_nheads: int, transformer_dim_feedforward: int, transformer_num_stages: int,
                 transformer_spatial_layers: int, transformer_temporal_layers: int, transformer_temporal_attn_type: str,
                 conv_dims: int, transformer_spatial_in_features: List[str], transformer_temporal_in_features: List[str],
                 num_clip_frames: int, cross_clip_training: bool, mfma_dtype: Optional[str] = None):
        super().__init__()
        self.transformer_temporal_layers = transformer_temporal_layers
        self.num_clip_frames = num_clip_frames
        sp = sorted(((k, v) for k, v in input_shape.items() if k in transformer_spatial_in_features), key=lambda kv: kv[1].stride)
        tp = sorted(((k, v) for k, v in input_shape.items() if k in transformer_temporal_in_features), key=lambda kv: kv[1].stride)
        self.transformer_spatial_in_features = [k for k, _ in sp]
        self.transformer_temporal_in_features = [k for k, _ in tp]
        chans = [v.channels for _, v in sp]
        self.transformer_num_spatial_feature_levels = len(sp)
        self.transformer_num_temporal_feature_levels = len(tp)
        ins = chans[::-1] if len(sp) > 1 else [chans[-1]]            # from low to high resolution (res5 -> res3)
        self.input_proj = nn.ModuleList([_conv_gn(c, conv_dims) for c in ins])
        self.output_proj = nn.ModuleList([_conv_gn(conv_dims, c) for c in ins])
        for proj in list(self.input_proj) + list(self.output_proj):
            nn.init.xavier_uniform_(proj[0].weight, gain=1)
            nn.init.constant_(proj[0].bias, 0)
        self.transformer = MSDeformAttnTransformerEncoderOnly(
            d_model=conv_dims, dropout=transformer_dropout, attn_drop=transformer_attn_drop, nhead=transformer_nheads,
            dim_feedforward=transformer_dim_feedforward, num_stages=transformer_num_stages,
            num_spatial_layers=transformer_spatial_layers, num_temporal_layers=transformer_temporal_layers,
            temporal_attn_type=transformer_temporal_attn_type, num_spatial_feature_levels=self.transformer_num_spatial_feature_levels,
            num_temporal_feature_levels=self.transformer_num_temporal_feature_levels)
        self.pe_layer = PositionEmbeddingSine(conv_dims // 2, normalize=True)
        self.pe_layer_3d = PositionEmbeddingSine3D(conv_dims // 2, normalize=True)
        self.cross_clip_training = cross_clip_training
        self.conv_dims = conv_dims
        self.mfma_dtype = mfma_dtype
        self._packed = None
        self._packed_key = None
        self._pos_cache = None          # (key, pos 2-D tokens, [pos 3-D per temporal level]) of the last shapes seen

    def _dtype(self) -> str:
        from . import modules
        return self.mfma_dtype or modules.default_operand_dtype()

    def _pack_projs(self):
        dt = self._dtype()
        from .modules import _param_key
        key = (_param_key(self.input_proj, dt), _param_key(self.output_proj, dt))     # (the cached walk: `.parameters()` costs ~10 us per projection)
        if self._packed is not None and key == self._packed_key:
            return self._packed
        projs = list(self.input_proj) + list(self.output_proj)
        L = _lib.lib()
        dev = self.input_proj[0][0].weight.device
        keep, bufs = [], []

        def f(t):
            tt = _dev_f32(t.detach(), "parameter")
            keep.append(tt)
            return tt.data_ptr()

        for m in projs:
            cout, cin = m[0].weight.shape[:2]
            ps = _lib.AxvsConvGnParams(f(m[0].weight), f(m[0].bias), f(m[1].weight), f(m[1].bias))
            buf = torch.empty(L.axvs_conv1x1_gn_packed_bytes(cin, cout), dtype=torch.uint8, device=dev)
            _lib.check(L.axvs_conv1x1_gn_pack(C.byref(ps), buf.data_ptr(), cin, cout, _lib.DTYPES[dt], _stream(dev)), "axvs_conv1x1_gn_pack")
            bufs.append(buf)
        torch.cuda.current_stream(dev).synchronize()
        n = len(self.input_proj)
        self._packed, self._packed_key = (bufs[:n], bufs[n:]), key
        return self._packed

    def forward_features(self, features):
        from .modules import _on
        with _on(next(iter(features.values())).device):
            return self._forward_features(features)

    def _forward_features_train(self, features):
        from .glue_training import conv_gn_train
        """train() mode (WC/msdeformattn.py:404-437, :91-174 under autograd): the 1x1 convolutions + GroupNorm run the library's training tier
        (round 6: axial_vs_amd.glue_training, forward and backward in HIP; rounds 3 - 5 used torch's kernels here), the level embeddings are torch parameters added by torch; the sine embeddings come from the library's kernels (constants); the encoder's layers run
        their training tiers (deformable attention: HIP forward / backward of the op; axial-trajectory layers: axvs_axial_layer_train_*)."""
        order = self.transformer_spatial_in_features[::-1]
        xs = [features[f] for f in order]
        BT = xs[0].shape[0]
        B = BT // self.num_clip_frames        # (self.training: WC/msdeformattn.py:406)
        T = BT // B
        dev = xs[0].device
        shapes = [(int(x.shape[2]), int(x.shape[3])) for x in xs]
        spatial = self.transformer.num_spatial_layers > 0
        srcs, poss, pos_3d = [], [], []
        for idx, (f, x) in enumerate(zip(order, xs)):
            H, W = shapes[idx]
            srcs.append(conv_gn_train(x, self.input_proj[idx][0], self.input_proj[idx][1], out_layout="tokens"))      # [BT, HW, Cd], in the library (round 6)
            if spatial:
                sine = torch.empty(BT, H * W, self.conv_dims, dtype=torch.float32, device=dev)
                self.pe_layer.tokens_into(sine, None, BT, H, W, 0)
                poss.append(sine + self.transformer.level_embed_2d[idx].view(1, 1, -1))
            if self.transformer_temporal_layers > 0 and f in self.transformer_temporal_in_features:
                sine3 = self.pe_layer_3d.channels_last(B, T, H, W, dev)
                pos_3d.append(torch.as_tensor(sine3) + self.transformer.level_embed_3d[len(pos_3d)].view(1, 1, 1, 1, -1))
        src = torch.cat(srcs, 1)
        if spatial:
            y, h_attn, w_attn = self.transformer.encoder(src, shapes, None, BT, torch.cat(poss, 1), None, pos_3d)
        else:
            y, h_attn, w_attn = self.transformer.encoder(src, shapes, pos_3d)
        out = {}
        for i, (f, z) in enumerate(zip(order, torch.split(y, [h * w for h, w in shapes], dim=1))):
            H, W = shapes[i]
            out[f] = conv_gn_train(z.contiguous(), self.output_proj[i][0], self.output_proj[i][1], out_layout="nchw", hw=(H, W))
        return out, h_attn, w_attn

    def _forward_features(self, features):
        if self.training:
            return self._forward_features_train(features)
        order = self.transformer_spatial_in_features[::-1]          # low -> high resolution (WC/msdeformattn.py:411)
        xs = [_dev_f32(features[f], f) for f in order]
        BT = xs[0].shape[0]
        B = BT // self.num_clip_frames if self.cross_clip_training else 1
        T = BT // B
        Cd = self.conv_dims
        dev = xs[0].device
        shapes = [(int(x.shape[2]), int(x.shape[3])) for x in xs]
        sizes = [h * w for h, w in shapes]
        S = sum(sizes)
        L = _lib.lib()
        dt = _lib.DTYPES[self._dtype()]
        st = _stream(dev)
        pin, pout = self._pack_projs()
        spatial = self.transformer.num_spatial_layers > 0
        src = torch.empty(BT, S, Cd, dtype=torch.float32, device=dev)
        wsb = max(L.axvs_conv1x1_gn_workspace_bytes(BT, hw, max(Cd, x.shape[1]), 32) for hw, x in zip(sizes, xs))
        ws = _workspace(dev, wsb)
        # The position embeddings (+ level embeddings) are functions of the shapes and of two small parameters: they are built once
        # per (shapes, parameter version) instead of once per forward (the reference recomputes them, WC/msdeformattn.py:104-118;
        # 7 launches / 55 us of a 1.3 ms forward at BASELINE config 3).  Nothing downstream writes into them.
        lvl2d_p = self.transformer.level_embed_2d if spatial else None       # (the temporal-only decoder has no 2-D level embedding)
        lvl3d_p = getattr(self.transformer, "level_embed_3d", None)
        pkey = (str(dev), B, T, tuple(shapes), spatial) + tuple(v for p_ in (lvl2d_p, lvl3d_p) if p_ is not None for v in (p_.data_ptr(), p_._version))
        cached = self._pos_cache if self._pos_cache is not None and self._pos_cache[0] == pkey else None
        pos, pos_3d = (cached[1], cached[2]) if cached else (torch.empty(BT, S, Cd, dtype=torch.float32, device=dev) if spatial else None, [])
        lvl2d = _dev_f32(lvl2d_p.detach(), "level_embed_2d") if spatial and not cached else None
        starts = [sum(sizes[:i]) for i in range(len(sizes))]

        def proj_in(idx):
            # (own stream, own scratch: the levels' projections are independent chains and run side by side)
            H, W = shapes[idx]
            wsi = _workspace(dev, wsb)
            _lib.check(L.axvs_conv1x1_gn_fwd(xs[idx].data_ptr(), 0, 0, 0, src.data_ptr() + starts[idx] * Cd * 4, 1, S * Cd, Cd, pin[idx].data_ptr(), BT,
                                             H * W, xs[idx].shape[1], Cd, 32, 1e-5, dt, wsi.data_ptr(), wsi.numel(), _stream(dev)), "axvs_conv1x1_gn_fwd")
            return ()
        _run_levels_concurrently([(lambda i=i: proj_in(i)) for i in range(len(xs))])
        row0 = 0
        for idx, (f, x) in enumerate(zip(order, xs)):
            H, W = shapes[idx]
            if not cached:
                if spatial:
                    self.pe_layer.tokens_into(pos, lvl2d[idx], BT, H, W, row0)
                if self.transformer_temporal_layers > 0 and f in self.transformer_temporal_in_features:
                    lvl3d = _dev_f32(lvl3d_p.detach(), "level_embed_3d")
                    pos_3d.append(self.pe_layer_3d.channels_last_with_level(B, T, H, W, lvl3d[len(pos_3d)]))
            row0 += H * W
        if not cached:
            self._pos_cache = (pkey, pos, pos_3d)
        if spatial:
            y, h_attn, w_attn = self.transformer.encoder(src, shapes, None, BT, pos, None, pos_3d)
        else:                                                           # temporal-only decoder (WC/msdeformattn.py:152-170)
            y, h_attn, w_attn = self.transformer.encoder(src, shapes, pos_3d)
        outs = [torch.empty(BT, xs[i].shape[1], shapes[i][0], shapes[i][1], dtype=torch.float32, device=dev) for i in range(len(order))]

        def proj_out(i):
            H, W = shapes[i]
            wsi = _workspace(dev, wsb)
            _lib.check(L.axvs_conv1x1_gn_fwd(y.data_ptr() + starts[i] * Cd * 4, 1, S * Cd, Cd, outs[i].data_ptr(), 0, 0, 0, pout[i].data_ptr(), BT, H * W,
                                             Cd, xs[i].shape[1], 32, 1e-5, dt, wsi.data_ptr(), wsi.numel(), _stream(dev)), "axvs_conv1x1_gn_fwd")
            return (outs[i],)
        _run_levels_concurrently([(lambda i=i: proj_out(i)) for i in range(len(order))])
        return {f: outs[i] for i, f in enumerate(order)}, h_attn, w_attn


class WithinClipTrackingModule(nn.Module):
    """WC/maxtron_within_clip_tracking_module.py:14-69 without the detectron2 decorators (register it with
    `SEM_SEG_HEADS_REGISTRY.register()(WithinClipTrackingModule)` where detectron2 is present; `from_config` reads the same keys)."""

    def __init__(self, input_shape, *, transformer_dropout: float, transformer_attn_drop: float, transformer_nheads: int,
                 transformer_dim_feedforward: int, transformer_num_stages: int, transformer_spatial_layers: int,
                 transformer_temporal_layers: int, transformer_temporal_attn_type: str, transformer_conv_dims: int,
                 transformer_spatial_in_features: List[str], transformer_temporal_in_features: List[str], num_clip_frames: int,
                 cross_clip_training: bool):
        super().__init__()
        self.within_clip_tracking_module = MSDeformAttnPixelDecoder(
            input_shape=input_shape, transformer_dropout=transformer_dropout, transformer_attn_drop=transformer_attn_drop,
            transformer_nheads=transformer_nheads, transformer_dim_feedforward=transformer_dim_feedforward,
            transformer_spatial_layers=transformer_spatial_layers, transformer_temporal_layers=transformer_temporal_layers,
            transformer_temporal_attn_type=transformer_temporal_attn_type, conv_dims=transformer_conv_dims,
            transformer_spatial_in_features=transformer_spatial_in_features,
            transformer_temporal_in_features=transformer_temporal_in_features, transformer_num_stages=transformer_num_stages,
            num_clip_frames=num_clip_frames, cross_clip_training=cross_clip_training)

    @classmethod
    def from_config(cls, cfg, input_shape):
        m = cfg.MODEL.MAXTRON.WITHIN_CLIP_TRACKING_MODULE
        return dict(input_shape={k: v for k, v in input_shape.items() if k in m.SPATIAL_IN_FEATURES},
                    transformer_dropout=m.DROPOUT, transformer_attn_drop=m.ATTN_DROP, transformer_nheads=m.NHEADS,
                    transformer_dim_feedforward=m.DIM_FEEDFORWARD, transformer_num_stages=m.NUM_STAGES,
                    transformer_spatial_layers=m.SPATIAL_LAYERS, transformer_temporal_layers=m.TEMPORAL_LAYERS,
                    transformer_temporal_attn_type=m.TEMPORAL_ATTN_TYPE, transformer_conv_dims=m.CONV_DIMS,
                    transformer_spatial_in_features=m.SPATIAL_IN_FEATURES, transformer_temporal_in_features=m.TEMPORAL_IN_FEATURES,
                    num_clip_frames=cfg.INPUT.NUM_CLIP_FRAMES, cross_clip_training=cfg.MODEL.MAXTRON.CROSS_CLIP_TRACKING_MODULE.ENABLE)

    def set_stack_precision(self, precision: str = "f16") -> "WithinClipTrackingModule":
        """Operand precision of the axial-trajectory (temporal) layers of the stack: 'f16' (default: 16-bit MFMA operands, every
        layer inside 1e-3 on its own, the free-running stack at 1.4e-3 max-norm / <= 1e-3 relative L2 on the temporal levels at
        BASELINE config 3) or 'f32' (the layers' fp32 tier: the stack then holds 1e-3 in max-norm too, at about 2.6x the time --
        the reference runs this stack in fp32 end to end, WC/msdeformattn.py:244-273).  Returns self."""
        if precision not in ("f16", "bf16", "f32", "f16+last_f32", "f16+final_f32"):
            raise ValueError(f"unknown precision {precision!r}")
        from .modules import TemporalAxialTrajectoryAttentionLayer, TemporalEncoder
        if precision == "f16+final_f32":      # fp32 tier for the last temporal layer of the LAST stage only
            encs = [m for m in self.modules() if isinstance(m, TemporalEncoder)]
            for e_i, m in enumerate(encs):
                layers = list(getattr(m, "temporal_layers", ()))
                for i, layer in enumerate(layers):
                    if isinstance(layer, TemporalAxialTrajectoryAttentionLayer):
                        layer.mfma_dtype = "f32" if (e_i == len(encs) - 1 and i == len(layers) - 1) else "f16"
            return self
        if precision == "f16+last_f32":
            # 16-bit operands except in the LAST temporal layer of every stage (the layer whose rounding errors no later layer's LayerNorm averages
            # out): measured in round 6 -- profiles/r6_stack_last_layer_f32.txt -- it does not bring the free-running stack inside 1e-3 max-norm
            for m in self.modules():
                if isinstance(m, TemporalEncoder):
                    layers = list(getattr(m, "temporal_layers", ()))
                    for i, layer in enumerate(layers):
                        if isinstance(layer, TemporalAxialTrajectoryAttentionLayer):
                            layer.mfma_dtype = "f32" if i == len(layers) - 1 else "f16"
            return self
        for m in self.modules():
            if isinstance(m, TemporalAxialTrajectoryAttentionLayer):
                m.mfma_dtype = precision
        return self

    def forward_features(self, features):
        within_clip_features, axial_height_attn, axial_width_attn = self.within_clip_tracking_module.forward_features(features)
        for k in within_clip_features:
            features[k] = within_clip_features[k]
        return features, axial_height_attn, axial_width_attn
