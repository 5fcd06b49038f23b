"""nn.Module mirror of the Tube-Link trajectory-attention plugin.

Reference: TL = MaXTron_Tube-Link/mmdet/models/plugins/msdeformattn_pixel_decoder.py
  `MultiScaleDeformableAxialTrajectoryAttention` (TL:393-638, registered as an mmcv ATTENTION plugin and built by the
  encoder's `BaseTransformerLayer`s): deformable sampling over all levels, then -- on the `num_temporal_levels` coarsest
  levels -- `f + gamma * TemporalEncoder(src=f, pos=query_pos3d[i])` on the sampled rows (TL:613-630, `gamma` TL:485-486),
  then `output_proj`, dropout and the identity shortcut (TL:633-638).

Same constructor keywords, attribute names and state-dict keys as the reference class, so `ATTENTION.register_module()`
can register this class under the reference's name and checkpoints load with strict=True.
All arithmetic runs in libaxvs.so: `axvs_msda_sample_fwd` -> `axvs_axial_layer_fwd` (TubeLinkTemporalEncoder) ->
`axvs_scaled_residual` -> `axvs_msda_output_proj_fwd`; PyTorch only slices the token buffer per level.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Optional

import torch
import torch.nn as nn
from torch import Tensor

from . import _lib
from .modules import TubeLinkTemporalEncoder, _dev_f32, _guarded, _param_key, _require_eval, _stream, _workspace
from .msda import _shapes_host


class MultiScaleDeformableAxialTrajectoryAttention(nn.Module):
    def __init__(self, embed_dims: int = 256, num_heads: int = 8, num_levels: int = 4, num_temporal_levels: int = 2,
                 num_temporal_layers: int = 1, num_temporal_dim: int = 1024, num_points: int = 4, im2col_step: int = 64,
                 dropout: float = 0.1, batch_first: bool = False, skip_connect: bool = True, attn_drop: float = 0.0,
                 norm_cfg: Optional[dict] = None, init_cfg=None, value_proj_ratio: float = 1.0,
                 mfma_dtype: Optional[str] = None):
        super().__init__()
        if embed_dims % num_heads != 0:
            raise ValueError(f'embed_dims must be divisible by num_heads, but got {embed_dims} and {num_heads}')     # TL:453-455
        if value_proj_ratio != 1.0:
            raise NotImplementedError("axial_vs_amd: value_proj_ratio != 1.0 has no HIP path (every shipped config uses 1.0)")
        self.norm_cfg = norm_cfg
        self.dropout = nn.Dropout(dropout)
        self.batch_first = batch_first
        self.im2col_step = im2col_step
        self.embed_dims = embed_dims
        self.num_levels = num_levels
        self.num_temporal_levels = num_temporal_levels
        self.num_temporal_layers = num_temporal_layers
        self.skip_connect = skip_connect
        self.attn_drop = attn_drop
        self.num_heads = num_heads
        self.num_points = num_points
        self.sampling_offsets = nn.Linear(embed_dims, num_heads * num_levels * num_points * 2)
        self.attention_weights = nn.Linear(embed_dims, num_heads * num_levels * num_points)
        self.value_proj = nn.Linear(embed_dims, embed_dims)
        self.output_proj = nn.Linear(embed_dims, embed_dims)
        # TL:484: TemporalEncoder(value_proj_size, num_temporal_dim, attn_drop=..., num_temporal_layer=...)
        self.temporal_layer = TubeLinkTemporalEncoder(embed_dims, num_temporal_dim, attn_drop=attn_drop,
                                                      num_temporal_layer=num_temporal_layers, n_heads=8, mfma_dtype=mfma_dtype)
        if self.skip_connect:
            self.gamma = nn.Parameter(1e-6 * torch.ones(embed_dims), requires_grad=True)
        self.mfma_dtype = mfma_dtype
        self._packed = None
        self._packed_key = None
        self.init_weights()

    def init_weights(self) -> None:
        """TL:489-507: zero offset weights, offset biases along num_heads directions with radius = point index, zero attention
        weights, xavier value / output projections."""
        nn.init.constant_(self.sampling_offsets.weight, 0.)
        thetas = torch.arange(self.num_heads, dtype=torch.float32) * (2.0 * math.pi / self.num_heads)
        grid = torch.stack([thetas.cos(), thetas.sin()], -1)
        grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(self.num_heads, 1, 1, 2).repeat(1, self.num_levels, self.num_points, 1)
        for i in range(self.num_points):
            grid[:, :, i, :] *= i + 1
        with torch.no_grad():
            self.sampling_offsets.bias.copy_(grid.view(-1))
        nn.init.constant_(self.attention_weights.weight, 0.)
        nn.init.constant_(self.attention_weights.bias, 0.)
        nn.init.xavier_uniform_(self.value_proj.weight)
        nn.init.constant_(self.value_proj.bias, 0.)
        nn.init.xavier_uniform_(self.output_proj.weight)
        nn.init.constant_(self.output_proj.bias, 0.)

    def _dtype(self) -> str:
        from . import modules
        return self.mfma_dtype or modules.default_operand_dtype()

    def _pack(self) -> Tensor:
        dt = self._dtype()
        key = tuple(_param_key(getattr(self, n), dt) for n in ("value_proj", "sampling_offsets", "attention_weights", "output_proj"))
        if self._packed is not None and key == self._packed_key:
            return self._packed
        L = _lib.lib()
        dev = self.value_proj.weight.device
        keep = []
        ps = _lib.AxvsMsdaParams()
        for name in ("value_proj", "sampling_offsets", "attention_weights", "output_proj"):
            lin = getattr(self, name)
            for suffix, t in (("_w", lin.weight), ("_b", lin.bias)):
                tt = _dev_f32(t.detach(), name)
                keep.append(tt)
                setattr(ps, name + suffix, tt.data_ptr())
        buf = torch.empty(L.axvs_msda_packed_bytes(self.embed_dims, self.num_heads, self.num_levels, self.num_points), dtype=torch.uint8,
                          device=dev)
        _lib.check(L.axvs_msda_pack(C.byref(ps), buf.data_ptr(), self.embed_dims, self.num_heads, self.num_levels, self.num_points,
                                    _lib.DTYPES[dt], _stream(dev)), "axvs_msda_pack")
        torch.cuda.current_stream(dev).synchronize()
        self._packed, self._packed_key = buf, key
        return buf

    @_guarded
    def forward(self, query: Tensor, key: Optional[Tensor] = None, value: Optional[Tensor] = None, identity: Optional[Tensor] = None,
                query_pos: Optional[Tensor] = None, query_pos3d: Optional[List[Tensor]] = None,
                key_padding_mask: Optional[Tensor] = None, reference_points: Optional[Tensor] = None,
                spatial_shapes=None, level_start_index=None, **kwargs) -> Tensor:
        """Same arguments as the reference (TL:511-565).  query / value / identity / query_pos: (num_query, bs, C), or
        (bs, num_query, C) with ``batch_first``; bs = B*T frames; query_pos3d[i]: [B, T, H_i, W_i, C] for the i-th (coarsest
        first) temporal level; reference_points (bs, num_query, num_levels, 2 | 4); spatial_shapes (num_levels, 2) = (h, w).
        Returns the tensor in the layout of `query`."""
        if self.training or (torch.is_grad_enabled() and query.requires_grad):
            return self._forward_autograd(query, value, identity, query_pos, query_pos3d, key_padding_mask, reference_points, spatial_shapes,
                                          level_start_index)
        v_is_q, id_is_q = value is None or value is query, identity is None or identity is query     # TL:567-570 defaults
        if value is None:
            value = query
        if identity is None:
            identity = query
        if not self.batch_first:                       # (num_query, bs, C) -> (bs, num_query, C)   TL:580-583
            query, value, identity = query.permute(1, 0, 2), value.permute(1, 0, 2), identity.permute(1, 0, 2)
            if query_pos is not None:
                query_pos = query_pos.permute(1, 0, 2)
        q = _dev_f32(query, "query")                   # (one layout copy for the three roles when they are the same tensor)
        v = q if v_is_q else _dev_f32(value, "value")
        ident = q if id_is_q else _dev_f32(identity, "identity")
        qp = _dev_f32(query_pos, "query_pos") if query_pos is not None else None
        ref = _dev_f32(reference_points, "reference_points")
        bs, nq, Cq = q.shape
        nv = v.shape[1]
        shp = _shapes_host(spatial_shapes)
        assert sum(h * w for h, w in shp) == nv                                                  # TL:587
        if ref.shape[-1] not in (2, 4):
            raise ValueError(f'Last dim of reference_points must be 2 or 4, but get {ref.shape[-1]} instead.')   # TL:603-606
        if Cq != self.embed_dims or len(shp) != self.num_levels or tuple(ref.shape[:3]) != (bs, nq, self.num_levels):
            raise RuntimeError(f"shape mismatch: query {tuple(q.shape)}, reference_points {tuple(ref.shape)}, levels {len(shp)}")
        if nq != nv:
            raise RuntimeError("the temporal section splits the output by level: num_query must equal sum(h*w) (TL:615-616)")
        mask = None
        if key_padding_mask is not None:
            if not key_padding_mask.is_cuda:
                raise RuntimeError("axial_vs_amd: key_padding_mask must be a CUDA tensor (no CPU fallback)")
            mask = key_padding_mask.to(torch.uint8).contiguous()
        L = _lib.lib()
        packed = self._pack()
        dev, st, dt = q.device, _stream(q.device), _lib.DTYPES[self._dtype()]
        ws = _workspace(dev, L.axvs_msda_workspace_bytes(bs, nq, nv, Cq, self.num_heads, self.num_levels, self.num_points))
        arr = (C.c_int * (2 * self.num_levels))(*[x for hw in shp for x in hw])
        sampled = torch.empty(bs, nq, Cq, dtype=torch.float32, device=dev)
        _lib.check(L.axvs_msda_sample_fwd(q.data_ptr(), qp.data_ptr() if qp is not None else None, ref.data_ptr(), ref.shape[-1],
                                          v.data_ptr(), mask.data_ptr() if mask is not None else None, arr, sampled.data_ptr(),
                                          packed.data_ptr(), bs, nq, nv, Cq, self.num_heads, self.num_levels, self.num_points, dt,
                                          ws.data_ptr(), ws.numel(), st), "axvs_msda_sample_fwd")
        # temporal section, coarsest levels first (TL:613-630); the other levels pass through
        start = 0
        for i, (h, w) in enumerate(shp[:self.num_temporal_levels]):
            f = sampled[:, start:start + h * w].contiguous()
            pos3d = query_pos3d[i]
            enc = self.temporal_layer(src=f, pos=pos3d)
            if self.skip_connect:
                g = _dev_f32(self.gamma.detach(), "gamma")
                _lib.check(L.axvs_scaled_residual(f.data_ptr(), enc.data_ptr(), g.data_ptr(), f.data_ptr(), f.numel(), Cq, st),
                           "axvs_scaled_residual")
                enc = f
            sampled[:, start:start + h * w] = enc
            start += h * w
        out = torch.empty_like(sampled)
        _lib.check(L.axvs_msda_output_proj_fwd(sampled.data_ptr(), ident.data_ptr(), out.data_ptr(), packed.data_ptr(), bs * nq, Cq,
                                               self.num_heads, self.num_levels, self.num_points, dt, st), "axvs_msda_output_proj_fwd")
        return out if self.batch_first else out.permute(1, 0, 2)

    def _forward_autograd(self, query, value, identity, query_pos, query_pos3d, key_padding_mask, reference_points, spatial_shapes,
                          level_start_index):
        """train() mode: the reference's forward (TL:561-638) under torch autograd -- deformable sampling (msda.deformable_sample: the
        op's HIP forward / backward inside), then per temporal level `f + gamma * encoder(f)` with the axial-trajectory layers on the
        library's training tier, output_proj, dropout, identity shortcut."""
        from .msda import _shapes_host, deformable_sample
        value = query if value is None else value
        shortcut = query if identity is None else identity
        q = query if query_pos is None else query + query_pos
        if not self.batch_first:
            q, value = q.permute(1, 0, 2), value.permute(1, 0, 2)
        shp = _shapes_host(spatial_shapes)
        assert sum(h * w for h, w in shp) == value.shape[1]
        sampled = deformable_sample(self.value_proj, self.sampling_offsets, self.attention_weights, q, value, reference_points, shp,
                                    level_start_index, key_padding_mask, self.num_heads, self.num_points, self.im2col_step)
        levels = list(torch.split(sampled, [h * w for h, w in shp], dim=1))
        for i in range(self.num_temporal_levels):                       # coarsest levels first (TL:613-630); the others pass through
            f = levels[i].contiguous()
            enc = self.temporal_layer(src=f, pos=query_pos3d[i])
            levels[i] = f + self.gamma * enc if self.skip_connect else enc
        out = self.output_proj(torch.cat(levels, dim=1))
        if not self.batch_first:
            out = out.permute(1, 0, 2)
        return self.dropout(out) + shortcut
