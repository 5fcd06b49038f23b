"""train() mode of the pixel decoder's 1x1 convolution + GroupNorm projections (WC/msdeformattn.py:349-375 under autograd): a
``torch.autograd.Function`` over libaxvs.so's ``axvs_conv1x1_gn_train_fwd`` / ``_bwd`` (round 6).  Nothing is computed here: forward and backward -- the projection
GEMM, the GroupNorm statistics and their gradients, the weight / bias / input gradients, the NCHW <-> token-row transposes -- run in the library.

    tokens = conv_gn_train(x_nchw, conv, gn, out_layout="tokens")      # input_proj:  [N, Cin, H, W]  ->  [N, H*W, Cout]
    maps   = conv_gn_train(tokens,  conv, gn, out_layout="nchw", hw=(H, W))   # output_proj: [N, H*W, Cin]  ->  [N, Cout, H, W]
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _lib


def _f32c(t: Tensor) -> Tensor:
    t = t.detach()
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


class _ConvGnTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, conv_w, conv_b, gn_w, gn_b, in_layout, out_layout, N, HW, Cin, Cout, groups, eps, hw):
        from .modules import _stream, _workspace
        if not x.is_cuda:
            raise RuntimeError("axial_vs_amd: the training tier needs GPU tensors; there is no CPU fallback")
        xs = _f32c(x)
        ws = [_f32c(w) for w in (conv_w, conv_b, gn_w, gn_b)]
        L = _lib.lib()
        dev = xs.device
        il, ol = (0 if in_layout == "nchw" else 1), (0 if out_layout == "nchw" else 1)
        ibs, ild = (0, 0) if il == 0 else (HW * Cin, Cin)
        obs, old = (0, 0) if ol == 0 else (HW * Cout, Cout)
        with torch.cuda.device(dev):
            out = torch.empty((N, Cout) + tuple(hw), dtype=torch.float32, device=dev) if ol == 0 else torch.empty(N, HW, Cout, dtype=torch.float32, device=dev)
            saved = torch.empty(max(L.axvs_conv1x1_gn_train_saved_bytes(N, HW, Cin, Cout, groups, il, ibs, ild), 1), dtype=torch.uint8, device=dev)
            scratch = _workspace(dev, L.axvs_conv1x1_gn_train_scratch_bytes(N, HW, Cin, Cout, groups, 1))
            ps = _lib.AxvsConvGnParams(*[w.data_ptr() for w in ws])
            _lib.check(L.axvs_conv1x1_gn_train_fwd(xs.data_ptr(), il, ibs, ild, out.data_ptr(), ol, obs, old, C.byref(ps), N, HW, Cin, Cout, groups, eps,
                                                   saved.data_ptr(), saved.numel(), scratch.data_ptr(), scratch.numel(), _stream(dev)), "axvs_conv1x1_gn_train_fwd")
        ctx.save_for_backward(xs, saved, *ws)
        ctx.cfg = (il, ol, ibs, ild, obs, old, N, HW, Cin, Cout, groups)
        ctx.in_dtypes = (x.dtype, conv_w.dtype, conv_b.dtype, gn_w.dtype, gn_b.dtype)
        ctx.x_shape = tuple(x.shape)
        return out

    @staticmethod
    def backward(ctx, d_out):
        from .modules import _stream, _workspace
        xs, saved, cw, cb, gw, gb = ctx.saved_tensors
        il, ol, ibs, ild, obs, old, N, HW, Cin, Cout, groups = ctx.cfg
        L = _lib.lib()
        dev = xs.device
        d = _f32c(d_out)
        with torch.cuda.device(dev):
            grads = [torch.empty_like(w) for w in (cw, cb, gw, gb)]
            need_dx = ctx.needs_input_grad[0]
            dx = torch.empty_like(xs) if need_dx else None
            scratch = _workspace(dev, L.axvs_conv1x1_gn_train_scratch_bytes(N, HW, Cin, Cout, groups, 1))
            ps = _lib.AxvsConvGnParams(cw.data_ptr(), cb.data_ptr(), gw.data_ptr(), gb.data_ptr())
            gs = _lib.AxvsConvGnParams(*[g.data_ptr() for g in grads])
            _lib.check(L.axvs_conv1x1_gn_train_bwd(d.data_ptr(), ol, obs, old, xs.data_ptr(), il, ibs, ild, C.byref(ps), C.byref(gs), dx.data_ptr() if need_dx else None,
                                                   N, HW, Cin, Cout, groups, saved.data_ptr(), saved.numel(), scratch.data_ptr(), scratch.numel(), _stream(dev)),
                       "axvs_conv1x1_gn_train_bwd")
        dts = ctx.in_dtypes
        out = [dx.reshape(ctx.x_shape).to(dts[0]) if need_dx else None] + [g.to(dt) for g, dt in zip(grads, dts[1:])]
        return tuple(out) + (None,) * 9


def conv_gn_train(x: Tensor, conv: torch.nn.Conv2d, gn: torch.nn.GroupNorm, out_layout: str = "nchw", hw: Optional[Tuple[int, int]] = None) -> Tensor:
    """GroupNorm(Conv2d 1x1 (x)) with autograd through the library.  x: [N, Cin, H, W] (NCHW) or [N, H*W, Cin] (token rows, contiguous; pass hw=(H, W) when the
    output is NCHW).  Returns [N, Cout, H, W] (out_layout="nchw") or [N, H*W, Cout] ("tokens")."""
    Cout, Cin = conv.weight.shape[:2]
    if conv.kernel_size != (1, 1) or conv.bias is None:
        raise NotImplementedError("axial_vs_amd: the projection is a 1x1 convolution with bias (WC/msdeformattn.py:353, :365)")
    if x.dim() == 4:
        N, H, W = x.shape[0], x.shape[2], x.shape[3]
        in_layout = "nchw"
    else:
        N = x.shape[0]
        H, W = hw if hw is not None else (x.shape[1], 1)
        in_layout = "tokens"
    return _ConvGnTrain.apply(x, conv.weight.reshape(Cout, Cin), conv.bias, gn.weight, gn.bias, in_layout, out_layout, N, H * W, Cin, Cout, gn.num_groups,
                              float(gn.eps), (H, W))
