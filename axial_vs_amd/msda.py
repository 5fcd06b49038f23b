"""nn.Module mirror of the multi-scale deformable attention of the within-clip pixel decoder.

Reference: OPS = MaXTron_Video-kMaX/maxtron_deeplab/modeling/within_clip_tracking_module/ops
  `MSDeformAttn` (OPS/modules/ms_deform_attn.py:35-125; Tube-Link uses mmcv's MultiScaleDeformableAttention with the same
  math), `MSDeformAttnFunction` / `ms_deform_attn_core_pytorch` (OPS/functions/ms_deform_attn_func.py:33-77).
Same constructor, parameter names (state-dict keys), initialisation and forward signature; forward runs in libaxvs.so.
eval(): the fused module kernels (`axvs_msda_fwd`, `axvs_msda_layer_fwd`).  train(): the reference module's own arithmetic under
torch autograd around `MSDeformAttnFunction`, whose forward AND backward are the HIP kernels of the native op
(`axvs_msda_core_fwd` / `axvs_msda_core_bwd` = the extension's ms_deform_attn_forward / _backward).  GPU only.
"""
from __future__ import annotations

import ctypes as C
import math
import warnings
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor

from . import _lib
from .modules import _dev_f32, _param_key, _require_eval, _stream, _workspace, _guarded


def _shapes_host(spatial_shapes) -> list:
    """(n_levels, 2) tensor / list of (H, W) -> host ints.  A device tensor costs one sync, like the reference's own
    `assert (...).sum() == Len_in` (OPS/modules/ms_deform_attn.py:96)."""
    if isinstance(spatial_shapes, Tensor):
        spatial_shapes = spatial_shapes.tolist()
    return [(int(h), int(w)) for h, w in spatial_shapes]


@_guarded
def ms_deform_attn_forward(value: Tensor, value_spatial_shapes, value_level_start_index, sampling_locations: Tensor,
                           attention_weights: Tensor, im2col_step: int = 64) -> Tensor:
    """Drop-in for `MSDA.ms_deform_attn_forward` / `MSDeformAttnFunction.apply` (forward):
    value [N,S,M,D], sampling_locations [N,Lq,M,L,P,2], attention_weights [N,Lq,M,L,P] -> [N,Lq,M*D].
    `value_level_start_index` and `im2col_step` are accepted for signature parity and not needed."""
    v = _dev_f32(value, "value")
    loc = _dev_f32(sampling_locations, "sampling_locations")
    aw = _dev_f32(attention_weights, "attention_weights")
    N, S, M, D = v.shape
    _, Lq, _, L, P, _ = loc.shape
    shp = _shapes_host(value_spatial_shapes)
    arr = (C.c_int * (2 * L))(*[x for hw in shp for x in hw])
    out = torch.empty(N, Lq, M * D, dtype=torch.float32, device=v.device)
    _lib.check(_lib.lib().axvs_msda_core_fwd(v.data_ptr(), arr, loc.data_ptr(), aw.data_ptr(), out.data_ptr(), N, S, M, D, Lq, L, P,
                                             _stream(v.device)), "axvs_msda_core_fwd")
    return out


@_guarded
def ms_deform_attn_backward(value: Tensor, value_spatial_shapes, value_level_start_index, sampling_locations: Tensor,
                            attention_weights: Tensor, grad_output: Tensor, im2col_step: int = 64):
    """Drop-in for `MSDA.ms_deform_attn_backward` (OPS/src/ms_deform_attn.h:49-67): -> (grad_value [N,S,M,D],
    grad_sampling_loc [N,Lq,M,L,P,2], grad_attn_weight [N,Lq,M,L,P]), fp32."""
    v = _dev_f32(value, "value")
    loc = _dev_f32(sampling_locations, "sampling_locations")
    aw = _dev_f32(attention_weights, "attention_weights")
    go = _dev_f32(grad_output, "grad_output")
    N, S, M, D = v.shape
    _, Lq, _, L, P, _ = loc.shape
    shp = _shapes_host(value_spatial_shapes)
    arr = (C.c_int * (2 * L))(*[x for hw in shp for x in hw])
    gv, gl, ga = torch.empty_like(v), torch.empty_like(loc), torch.empty_like(aw)
    _lib.check(_lib.lib().axvs_msda_core_bwd(v.data_ptr(), arr, loc.data_ptr(), aw.data_ptr(), go.data_ptr(), gv.data_ptr(), gl.data_ptr(),
                                             ga.data_ptr(), N, S, M, D, Lq, L, P, _stream(v.device)), "axvs_msda_core_bwd")
    return gv, gl, ga


class MSDeformAttnFunction(torch.autograd.Function):
    """The reference's autograd wrapper of its CUDA op (OPS/functions/ms_deform_attn_func.py:32-52) over the HIP kernels: same
    signature, same saved tensors; gradients for value, sampling_locations and attention_weights."""

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights, im2col_step):
        ctx.im2col_step = im2col_step
        ctx.shapes = _shapes_host(value_spatial_shapes)
        ctx.in_dtypes = (value.dtype, sampling_locations.dtype, attention_weights.dtype)
        v, loc, aw = (t.detach().float().contiguous() for t in (value, sampling_locations, attention_weights))
        output = ms_deform_attn_forward(v, ctx.shapes, value_level_start_index, loc, aw, im2col_step)
        ctx.save_for_backward(v, loc, aw)
        return output

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_output):
        v, loc, aw = ctx.saved_tensors
        gv, gl, ga = ms_deform_attn_backward(v, ctx.shapes, None, loc, aw, grad_output.float().contiguous(), ctx.im2col_step)
        dv, dl, da = ctx.in_dtypes
        return gv.to(dv), None, None, gl.to(dl), ga.to(da), None


class MSDeformAttn(nn.Module):
    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4, mfma_dtype: Optional[str] = None):
        super().__init__()
        if d_model % n_heads != 0:
            raise ValueError('d_model must be divisible by n_heads, but got {} and {}'.format(d_model, n_heads))
        d = d_model // n_heads
        if d & (d - 1):
            warnings.warn("d_model / n_heads is not a power of 2 (the reference warns about this too)")
        self.im2col_step = 128
        self.d_model, self.n_levels, self.n_heads, self.n_points = d_model, n_levels, n_heads, n_points
        self.sampling_offsets = nn.Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = nn.Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = nn.Linear(d_model, d_model)
        self.output_proj = nn.Linear(d_model, d_model)
        self.mfma_dtype = mfma_dtype
        self._packed = None
        self._packed_key = None
        self._reset_parameters()

    def _reset_parameters(self):
        """OPS/modules/ms_deform_attn.py:66-79: zero offset/attention weights, offsets biased along n_heads directions with
        radius growing with the point index, xavier value/output projections."""
        nn.init.constant_(self.sampling_offsets.weight.data, 0.)
        ang = torch.arange(self.n_heads, dtype=torch.float32) * (2.0 * math.pi / self.n_heads)
        dirs = torch.stack([ang.cos(), ang.sin()], -1)
        dirs = dirs / dirs.abs().max(-1, keepdim=True)[0]
        grid = dirs.view(self.n_heads, 1, 1, 2).repeat(1, self.n_levels, self.n_points, 1)
        grid = grid * torch.arange(1, self.n_points + 1, dtype=torch.float32).view(1, 1, self.n_points, 1)
        with torch.no_grad():
            self.sampling_offsets.bias = nn.Parameter(grid.reshape(-1))
        nn.init.constant_(self.attention_weights.weight.data, 0.)
        nn.init.constant_(self.attention_weights.bias.data, 0.)
        nn.init.xavier_uniform_(self.value_proj.weight.data)
        nn.init.constant_(self.value_proj.bias.data, 0.)
        nn.init.xavier_uniform_(self.output_proj.weight.data)
        nn.init.constant_(self.output_proj.bias.data, 0.)

    def _dtype(self) -> str:
        from . import modules
        return self.mfma_dtype or modules.default_operand_dtype()

    def _pack(self):
        dt = self._dtype()
        key = _param_key(self, dt)
        if self._packed is not None and key == self._packed_key:
            return self._packed
        L = _lib.lib()
        dev = self.value_proj.weight.device
        keep = []

        def f(t):
            tt = _dev_f32(t.detach(), "parameter")
            keep.append(tt)
            return tt.data_ptr()

        ps = _lib.AxvsMsdaParams()
        for name in ("value_proj", "sampling_offsets", "attention_weights", "output_proj"):
            lin = getattr(self, name)
            setattr(ps, name + "_w", f(lin.weight))
            setattr(ps, name + "_b", f(lin.bias))
        buf = torch.empty(L.axvs_msda_packed_bytes(self.d_model, self.n_heads, self.n_levels, self.n_points), dtype=torch.uint8, device=dev)
        _lib.check(L.axvs_msda_pack(C.byref(ps), buf.data_ptr(), self.d_model, self.n_heads, self.n_levels, self.n_points,
                                    _lib.DTYPES[dt], _stream(dev)), "axvs_msda_pack")
        torch.cuda.current_stream(dev).synchronize()
        self._packed, self._packed_key = buf, key
        return buf

    @_guarded
    def forward(self, query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index=None,
                input_padding_mask=None):
        """query (N, Len_q, C); reference_points (N, Len_q, n_levels, 2 | 4) in [0,1]; input_flatten (N, sum H_l W_l, C);
        input_spatial_shapes (n_levels, 2) = (H_l, W_l); input_padding_mask (N, sum H_l W_l) True = padding -> (N, Len_q, C)"""
        if self.training or (torch.is_grad_enabled() and any(t.requires_grad for t in (query, input_flatten))):
            return self._forward_autograd(query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index,
                                          input_padding_mask)
        q = _dev_f32(query, "query")
        x = _dev_f32(input_flatten, "input_flatten")
        ref = _dev_f32(reference_points, "reference_points")
        N, Lq, Cq = q.shape
        S = x.shape[1]
        shp = _shapes_host(input_spatial_shapes)
        if sum(h * w for h, w in shp) != S:
            raise AssertionError("input_spatial_shapes do not cover input_flatten")          # reference: assert, :96
        if ref.shape[-1] not in (2, 4):
            raise ValueError('Last dim of reference_points must be 2 or 4, but get {} instead.'.format(ref.shape[-1]))
        if Cq != self.d_model or len(shp) != self.n_levels or ref.shape[:3] != (N, Lq, self.n_levels):
            raise RuntimeError(f"shape mismatch: query {tuple(q.shape)}, reference_points {tuple(ref.shape)}, levels {len(shp)}")
        mask = None
        if input_padding_mask is not None:
            if not input_padding_mask.is_cuda:
                raise RuntimeError("axial_vs_amd: input_padding_mask must be a CUDA tensor (no CPU fallback)")
            mask = input_padding_mask.to(torch.uint8).contiguous()
        L = _lib.lib()
        packed = self._pack()
        dev = q.device
        ws = _workspace(dev, L.axvs_msda_workspace_bytes(N, Lq, S, self.d_model, self.n_heads, self.n_levels, self.n_points))
        arr = (C.c_int * (2 * self.n_levels))(*[v for hw in shp for v in hw])
        out = torch.empty(N, Lq, self.d_model, dtype=torch.float32, device=dev)
        _lib.check(L.axvs_msda_fwd(q.data_ptr(), ref.data_ptr(), ref.shape[-1], x.data_ptr(), mask.data_ptr() if mask is not None else None,
                                   arr, out.data_ptr(), packed.data_ptr(), N, Lq, S, self.d_model, self.n_heads, self.n_levels,
                                   self.n_points, _lib.DTYPES[self._dtype()], ws.data_ptr(), ws.numel(), _stream(dev)), "axvs_msda_fwd")
        return out


    def _forward_autograd(self, query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index, input_padding_mask):
        """train() mode (or gradients wanted): the arithmetic of the reference module (OPS/modules/ms_deform_attn.py:93-125) under
        torch autograd -- the four nn.Linear layers, the softmax over the L*P logits, the sampling locations -- around
        `MSDeformAttnFunction`, whose forward and backward are the HIP kernels of the native op (axvs_msda_core_fwd / _bwd)."""
        shp = _shapes_host(input_spatial_shapes)
        if sum(h * w for h, w in shp) != input_flatten.shape[1]:
            raise AssertionError("input_spatial_shapes do not cover input_flatten")
        sampled = deformable_sample(self.value_proj, self.sampling_offsets, self.attention_weights, query, input_flatten, reference_points, shp,
                                    input_level_start_index, input_padding_mask, self.n_heads, self.n_points, self.im2col_step)
        return self.output_proj(sampled)


def sampling_locations(reference_points: Tensor, offsets: Tensor, shp, n_points: int) -> Tensor:
    """OPS/modules/ms_deform_attn.py:107-117: reference points [N,Lq,L,2] + offsets in pixels of each level, or reference boxes
    [N,Lq,L,4] + offsets in units of half a box per point.  offsets [N,Lq,M,L,P,2] -> locations of the same shape, in [0,1]."""
    ref = reference_points[:, :, None, :, None, :]
    if reference_points.shape[-1] == 2:
        wh = torch.tensor([[w, h] for h, w in shp], dtype=offsets.dtype, device=offsets.device)
        return ref + offsets / wh[None, None, None, :, None, :]
    if reference_points.shape[-1] == 4:
        return ref[..., :2] + offsets / n_points * ref[..., 2:] * 0.5
    raise ValueError('Last dim of reference_points must be 2 or 4, but get {} instead.'.format(reference_points.shape[-1]))


def deformable_sample(value_proj, offsets_proj, weights_proj, query, value_in, reference_points, shp, level_start_index, padding_mask,
                      n_heads: int, n_points: int, im2col_step: int) -> Tensor:
    """The part of MSDeformAttn.forward in front of output_proj, differentiable: value_proj (+ padding mask) -> [N,S,M,D]; offsets and
    softmaxed weights from the query; `MSDeformAttnFunction` on the sampling locations.  Shared by the Video-kMaX module and the
    Tube-Link plugin (same math: TL/mmdet/models/plugins/msdeformattn_pixel_decoder.py:589-611)."""
    N, Lq, _ = query.shape
    S, L = value_in.shape[1], len(shp)
    value = value_proj(value_in)
    if padding_mask is not None:
        value = value.masked_fill(padding_mask[..., None], 0.0)
    value = value.view(N, S, n_heads, -1)
    offsets = offsets_proj(query).view(N, Lq, n_heads, L, n_points, 2)
    weights = F.softmax(weights_proj(query).view(N, Lq, n_heads, L * n_points), -1).view(N, Lq, n_heads, L, n_points)
    return MSDeformAttnFunction.apply(value, shp, level_start_index, sampling_locations(reference_points, offsets, shp, n_points), weights,
                                      im2col_step)


class MSDeformAttnTransformerEncoderLayer(nn.Module):
    """Mirror of WC/msdeformattn.py:177-216 (the spatial layer of every within-clip stage): deformable self-attention +
    residual, norm1, FFN, norm2 -- one C-ABI call (`axvs_msda_layer_fwd`); the FFN half is the axial layer's fused kernel."""

    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_levels=4, n_heads=8, n_points=4,
                 mfma_dtype: Optional[str] = None):
        super().__init__()
        if activation != "relu":
            if activation in ("gelu", "glu"):
                raise NotImplementedError("axial_vs_amd: only activation='relu' (every shipped config) has a HIP path")
            raise RuntimeError(f"activation should be relu/gelu, not {activation}.")
        self.self_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.dropout2 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.dropout3 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)
        self.d_model, self.d_ffn = d_model, d_ffn
        self.mfma_dtype = mfma_dtype
        self._packed = None
        self._packed_key = None

    @staticmethod
    def with_pos_embed(tensor, pos):
        return tensor if pos is None else tensor + pos

    def _dtype(self) -> str:
        from . import modules
        return self.mfma_dtype or modules.default_operand_dtype()

    def _pack(self):
        dt = self._dtype()
        key = _param_key(self, dt)
        if self._packed is not None and key == self._packed_key:
            return self._packed
        L = _lib.lib()
        a = self.self_attn
        dev = self.norm1.weight.device
        keep = []

        def f(t):
            tt = _dev_f32(t.detach(), "parameter")
            keep.append(tt)
            return tt.data_ptr()

        ps = _lib.AxvsMsdaLayerParams()
        for name in ("value_proj", "sampling_offsets", "attention_weights", "output_proj"):
            lin = getattr(a, name)
            setattr(ps.self_attn, name + "_w", f(lin.weight))
            setattr(ps.self_attn, name + "_b", f(lin.bias))
        for name in ("norm1", "linear1", "linear2", "norm2"):
            mod = getattr(self, name)
            setattr(ps, name + "_w", f(mod.weight))
            setattr(ps, name + "_b", f(mod.bias))
        buf = torch.empty(L.axvs_msda_layer_packed_bytes(self.d_model, a.n_heads, a.n_levels, a.n_points, self.d_ffn), dtype=torch.uint8,
                          device=dev)
        _lib.check(L.axvs_msda_layer_pack(C.byref(ps), buf.data_ptr(), self.d_model, a.n_heads, a.n_levels, a.n_points, self.d_ffn,
                                          _lib.DTYPES[dt], _stream(dev)), "axvs_msda_layer_pack")
        torch.cuda.current_stream(dev).synchronize()
        self._packed, self._packed_key = buf, key
        return buf

    @_guarded
    def forward(self, src, pos, reference_points, spatial_shapes, level_start_index=None, padding_mask=None):
        if self.training or (torch.is_grad_enabled() and src.requires_grad):
            # train() mode: the reference layer's forward (WC/msdeformattn.py:203-216) under torch autograd; the deformable attention
            # op inside runs (forward and backward) on the HIP kernels through MSDeformAttnFunction
            src2 = self.self_attn(self.with_pos_embed(src, pos), reference_points, src, spatial_shapes, level_start_index, padding_mask)
            src = self.norm1(src + self.dropout1(src2))
            src2 = self.linear2(self.dropout2(F.relu(self.linear1(src))))
            return self.norm2(src + self.dropout3(src2))
        x = _dev_f32(src, "src")
        p = _dev_f32(pos, "pos") if pos is not None else None
        ref = _dev_f32(reference_points, "reference_points")
        a = self.self_attn
        N, S, Cq = x.shape
        shp = _shapes_host(spatial_shapes)
        if sum(h * w for h, w in shp) != S:
            raise AssertionError("spatial_shapes do not cover src")
        if ref.shape[-1] not in (2, 4):
            raise ValueError('Last dim of reference_points must be 2 or 4, but get {} instead.'.format(ref.shape[-1]))
        if Cq != self.d_model or len(shp) != a.n_levels or ref.shape[:3] != (N, S, a.n_levels) or (p is not None and p.shape != x.shape):
            raise RuntimeError(f"shape mismatch: src {tuple(x.shape)}, reference_points {tuple(ref.shape)}, levels {len(shp)}")
        mask = None
        if padding_mask is not None:
            if not padding_mask.is_cuda:
                raise RuntimeError("axial_vs_amd: padding_mask must be a CUDA tensor (no CPU fallback)")
            mask = padding_mask.to(torch.uint8).contiguous()
        L = _lib.lib()
        packed = self._pack()
        dev = x.device
        ws = _workspace(dev, L.axvs_msda_layer_workspace_bytes(N, S, self.d_model, a.n_heads, a.n_levels, a.n_points, self.d_ffn))
        arr = (C.c_int * (2 * a.n_levels))(*[v for hw in shp for v in hw])
        out = torch.empty_like(x)
        _lib.check(L.axvs_msda_layer_fwd(x.data_ptr(), p.data_ptr() if p is not None else None, ref.data_ptr(), ref.shape[-1],
                                         mask.data_ptr() if mask is not None else None, arr, out.data_ptr(), packed.data_ptr(), N, S,
                                         self.d_model, a.n_heads, a.n_levels, a.n_points, self.d_ffn, _lib.DTYPES[self._dtype()],
                                         ws.data_ptr(), ws.numel(), _stream(dev)), "axvs_msda_layer_fwd")
        return out
