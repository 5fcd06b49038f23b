"""Training path of TemporalAxialTrajectoryAttentionLayer: autograd over libaxvs.so's training tier.

Reference: the layer in ``train()`` mode under autograd, WC/temporal_attention.py:187-220 (and TrajectoryAttention :35-76), as the
shipped configs train it (``ATTN_DROP: 0.1``, AMP -- VK/configs/VIPSeg/.../maxtron_wc_convnext_large.yaml).  The forward and the
backward pass both run in the library (``axvs_axial_layer_train_fwd`` / ``_bwd``: fp32 activations, hand-written attention /
softmax / dropout / LayerNorm kernels, split-precision bf16 MFMA GEMMs for the Linear layers); this file is the ``torch.autograd.Function`` that
binds them, nothing is computed here.

* ``layer.recompute = False`` (default): the activations stay in HBM between forward and backward (~44 C floats per token and layer),
  like the reference under autograd.  ``layer.recompute = True``: the forward pass keeps only (src, pos, seed); backward rebuilds the
  activations first (one more forward, no memory held).
* dropout masks are a counter-based hash of (seed, site, element offset) -- see include/axvs.h -- so they are regenerated, never
  stored; ``seed`` comes from torch's CPU generator (``torch.manual_seed`` makes runs repeatable) or ``layer.dropout_seed``.
* AMP: under ``torch.autocast`` the inputs are cast to fp32 at the boundary and the layer returns fp32 (LayerNorm output is fp32
  under autocast in the reference as well); gradients come back in each input's own dtype, so ``GradScaler`` works unchanged.  The
  Linear layers then multiply ONE 16-bit piece per operand in the autocast dtype (bf16 / fp16, fp32 accumulation) -- what autocast
  gives the reference's ``nn.Linear`` -- in the forward, input-gradient and weight-gradient GEMMs (library option ``train_amp``);
  ``layer.amp_compute = False`` keeps the split-precision (fp32-accurate) products.  Attention, softmax, LayerNorm and bias
  gradients stay fp32.
"""
from __future__ import annotations

import ctypes as C
from typing import List

import torch
from torch import Tensor

from . import _lib

_TRAJ = ("q", "k", "v", "proj_q", "proj_kv", "proj")
_TAIL = ("norm1", "linear1", "linear2", "norm2")


def layer_parameters(layer) -> List[Tensor]:
    """The layer's parameters in AxvsAxialLayerParams field order (include/axvs.h)."""
    ps: List[Tensor] = []
    for attn in (layer.height_attn, layer.width_attn):
        for n in _TRAJ:
            m = getattr(attn, n)
            ps += [m.weight, m.bias]
    for n in _TAIL:
        m = getattr(layer, n)
        ps += [m.weight, m.bias]
    return ps


def _struct(ptrs: List[int]) -> _lib.AxvsAxialLayerParams:
    s = _lib.AxvsAxialLayerParams()
    s.height_attn = _lib.AxvsTrajParams(*ptrs[0:12])
    s.width_attn = _lib.AxvsTrajParams(*ptrs[12:24])
    for name, p in zip(("norm1_w", "norm1_b", "linear1_w", "linear1_b", "linear2_w", "linear2_b", "norm2_w", "norm2_b"), ptrs[24:32]):
        setattr(s, name, p)
    return s


def _f32c(t: Tensor) -> Tensor:
    t = t.detach()
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


class _AxialLayerTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, pos, dims, p_dropout, p_attn_drop, seed, recompute, *params):
        from .modules import _stream, _workspace
        B, T, H, W, C_, heads, F = dims
        if not src.is_cuda:
            raise RuntimeError("axial_vs_amd: the training tier needs GPU tensors; there is no CPU fallback")
        s, p = _f32c(src), _f32c(pos)
        ws = [_f32c(w) for w in params]
        L = _lib.lib()
        dev = s.device
        nsaved = L.axvs_axial_layer_train_saved_bytes(B, T, H, W, C_, heads, F)
        if nsaved == 0:
            raise RuntimeError("axvs_axial_layer_train_saved_bytes: " + L.axvs_last_error().decode())
        with torch.cuda.device(dev):
            out = torch.empty_like(s)
            # with recompute the forward's activations are scratch too: they live in the shared workspace, after the scratch part
            nscr = L.axvs_axial_layer_train_scratch_bytes(B, T, H, W, C_, heads, F, 0)
            if recompute:
                buf = _workspace(dev, nscr + nsaved)
                scratch_ptr, saved_ptr, saved = buf.data_ptr(), buf.data_ptr() + nscr, None
            else:
                saved = torch.empty(nsaved, dtype=torch.uint8, device=dev)
                buf = _workspace(dev, nscr)
                scratch_ptr, saved_ptr = buf.data_ptr(), saved.data_ptr()
            st = _struct([w.data_ptr() for w in ws])
            _lib.check(L.axvs_axial_layer_train_fwd(s.data_ptr(), p.data_ptr(), out.data_ptr(), C.byref(st), B, T, H, W, C_, heads, F,
                                                    float(p_dropout), float(p_attn_drop), int(seed), saved_ptr, nsaved, scratch_ptr, nscr,
                                                    _stream(dev)), "axvs_axial_layer_train_fwd")
        ctx.save_for_backward(s, p, *ws)
        ctx.amp = _lib.current_amp()
        ctx.cfg = (dims, float(p_dropout), float(p_attn_drop), int(seed), bool(recompute))
        ctx.saved_buf = saved
        ctx.in_dtypes = (src.dtype, pos.dtype, [w.dtype for w in params])
        ctx.shapes = (src.shape, pos.shape)
        return out.view(src.shape)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_out):
        from .modules import _stream, _workspace
        s, p, *ws = ctx.saved_tensors
        dims, p_dropout, p_attn_drop, seed, recompute = ctx.cfg
        B, T, H, W, C_, heads, F = dims
        L = _lib.lib()
        dev = s.device
        with torch.cuda.device(dev):
            g = _f32c(d_out)
            d_src = torch.empty_like(s)
            want_pos = ctx.needs_input_grad[1]
            d_pos = torch.empty_like(p) if want_pos else None
            sizes = [w.numel() for w in ws]
            flat = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
            grads, off = [], 0
            for w, n in zip(ws, sizes):
                grads.append(flat[off:off + n].view(w.shape))
                off += n
            nsaved = L.axvs_axial_layer_train_saved_bytes(B, T, H, W, C_, heads, F)
            nscr = L.axvs_axial_layer_train_scratch_bytes(B, T, H, W, C_, heads, F, 1)
            if recompute:
                buf = _workspace(dev, nscr + nsaved)
                scratch_ptr, saved_ptr = buf.data_ptr(), buf.data_ptr() + nscr
            else:
                buf = _workspace(dev, nscr)
                scratch_ptr, saved_ptr = buf.data_ptr(), ctx.saved_buf.data_ptr()
            st = _struct([w.data_ptr() for w in ws])
            gs = _struct([t.data_ptr() for t in grads])
            with _lib.train_amp(ctx.amp):
                _lib.check(L.axvs_axial_layer_train_bwd(g.data_ptr(), s.data_ptr(), p.data_ptr(), C.byref(st), C.byref(gs), d_src.data_ptr(),
                                                        d_pos.data_ptr() if want_pos else None, B, T, H, W, C_, heads, F, p_dropout, p_attn_drop,
                                                        seed, int(recompute), saved_ptr, nsaved, scratch_ptr, nscr, _stream(dev)),
                           "axvs_axial_layer_train_bwd")
        # (the saved activations stay with ctx until autograd releases it: a second backward through the same graph --
        #  retain_graph=True, shared subgraphs -- finds them again)
        sd, pd, wd = ctx.in_dtypes
        out_grads = [gr.to(dt) for gr, dt in zip(grads, wd)]
        return (d_src.view(ctx.shapes[0]).to(sd), d_pos.view(ctx.shapes[1]).to(pd) if want_pos else None, None, None, None, None, None,
                *out_grads)


def axial_layer_train(layer, src: Tensor, pos: Tensor, dropout: bool = True, recompute: bool = True) -> Tensor:
    """Differentiable forward of a TemporalAxialTrajectoryAttentionLayer through the training tier.
    src [(B T),(H W),C], pos [B,T,H,W,C] -> out like src (fp32).  dropout=False: probabilities forced to 0 (gradients in eval mode)."""
    if layer.activation != "relu":
        raise NotImplementedError("axial_vs_amd: only activation='relu' (every shipped config) has a HIP path")
    if abs(layer.norm1.eps - 1e-5) > 0 or abs(layer.norm2.eps - 1e-5) > 0:
        raise NotImplementedError("axial_vs_amd: LayerNorm eps must be 1e-5")
    B, T, H, W = pos.shape[:4]
    C_ = src.shape[-1]
    if src.numel() != B * T * H * W * C_ or pos.shape[-1] != C_:
        raise RuntimeError(f"src {tuple(src.shape)} does not match pos {tuple(pos.shape)}")
    p_drop = float(layer.dropout2.p) if dropout else 0.0        # = the attention maps' dropout (reference :164-165) = dropout2 = dropout3
    p_attn = float(layer.dropout1.p) if dropout else 0.0
    seed = getattr(layer, "dropout_seed", None)
    if seed is None:
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if (p_drop > 0 or p_attn > 0) else 0
    dims = (int(B), int(T), int(H), int(W), int(C_), int(layer.n_heads), int(layer.linear1.out_features))
    args = (src, pos, dims, p_drop, p_attn, int(seed), bool(recompute), *layer_parameters(layer))
    if torch.is_autocast_enabled():
        amp = _lib.autocast_mode(layer)       # (read before autocast is switched off for the call)
        with torch.autocast(device_type="cuda", enabled=False), _lib.train_amp(amp):
            return _AxialLayerTrain.apply(*args)
    return _AxialLayerTrain.apply(*args)
