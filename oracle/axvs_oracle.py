"""CPU oracle for the axial-trajectory-attention hot path.  TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it.  The shipped path
(``axial_vs_amd``) never imports anything under ``oracle/`` and fails loudly when the
HIP library is missing.

It is an independent restatement (plain torch CPU ops, explicit per-frame loops, no
einops) of the reference algorithm.  Every function cites the reference lines it
follows, with the shorthands

    WC = MaXTron_Video-kMaX/maxtron_deeplab/modeling/within_clip_tracking_module
    CC = MaXTron_Video-kMaX/maxtron_deeplab/modeling/cross_clip_tracking_module
    TL = MaXTron_Tube-Link

Parity pinning: the reference has no tests or golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference
itself: ``oracle/gen_golden.py`` imports the reference modules from /root/reference
in the build container and stores input/weight seeds + outputs in ``tests/golden``;
``tests/test_oracle_golden.py`` replays them.

All functions are dtype-generic: run them in float64 to obtain a reference that is
tighter than the fp32 PyTorch path itself.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Weights = Dict[str, Tensor]


# --------------------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------------------
def _linear(x: Tensor, w: Weights, name: str) -> Tensor:
    """y = x W^T + b with nn.Linear's [out, in] weight layout."""
    y = x @ w[name + ".weight"].to(x.dtype).t()
    b = w.get(name + ".bias")
    return y if b is None else y + b.to(x.dtype)


def _layer_norm(x: Tensor, w: Weights, name: str, eps: float = 1e-5) -> Tensor:
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w[name + ".weight"].to(x.dtype) + w[name + ".bias"].to(x.dtype)


def _sub(w: Weights, prefix: str) -> Weights:
    p = prefix + "."
    return {k[len(p):]: v for k, v in w.items() if k.startswith(p)}


# --------------------------------------------------------------------------------------
# (a2)+(a3) trajectory attention, q/k/v flavour (WC) and fused-qkv flavour (CC)
# --------------------------------------------------------------------------------------
def _trajectory_core(q: Tensor, k: Tensor, v: Tensor, w: Weights, num_frames: int, heads: int,
                     want_attn: bool, attn_keep: Optional[Tensor] = None) -> Tuple[Tensor, Optional[Tensor]]:
    """Shared by both flavours once q,k,v [S, N, C] are projected.

    Spatial half   WC/temporal_attention.py:46-57   (CC/...:103-113)
    Temporal half  WC/temporal_attention.py:60-75   (CC/...:116-129)
    """
    S, N, C = q.shape
    T = num_frames
    L = N // T
    assert L * T == N, "tokens per sequence must be num_frames * axis_len"
    d = C // heads
    scale = d ** -0.5

    qh = q.reshape(S, N, heads, d).permute(0, 2, 1, 3)            # [S,h,N,d]
    kh = k.reshape(S, N, heads, d).permute(0, 2, 1, 3)
    vh = v.reshape(S, N, heads, d).permute(0, 2, 1, 3)

    # spatial half: one softmax per (query, frame) over that frame's L keys
    x = q.new_empty(S, N, T, C)                                   # x[s, q, f, (h d)]
    attn_maps = q.new_empty(S, heads, N, T, L) if want_attn else None
    for f in range(T):
        kf = kh[:, :, f * L:(f + 1) * L]                          # [S,h,L,d]
        vf = vh[:, :, f * L:(f + 1) * L]
        logits = torch.matmul(qh, kf.transpose(-1, -2)) * scale   # [S,h,N,L]
        p = torch.softmax(logits, dim=-1)
        if want_attn:
            attn_maps[:, :, :, f] = p
        if attn_keep is not None:                                 # train mode: self.attn_drop(space_attn), :55
            p = p * attn_keep[:, :, :, f]
        xf = torch.matmul(p, vf)                                  # [S,h,N,d]
        x[:, :, f] = xf.permute(0, 2, 1, 3).reshape(S, N, C)

    # temporal half: the query is the token's own-frame slot of x (the "diagonal")
    own = torch.arange(N) // L                                    # frame of token n
    x_diag = x[:, torch.arange(N), own]                           # [S,N,C]
    q2 = _linear(x_diag, w, "proj_q") * scale                     # [S,N,C]
    kv2 = _linear(x, w, "proj_kv")                                # [S,N,T,2C]
    k2, v2 = kv2[..., :C], kv2[..., C:]
    q2h = q2.reshape(S, N, heads, d)
    k2h = k2.reshape(S, N, T, heads, d)
    v2h = v2.reshape(S, N, T, heads, d)
    tl = (q2h.unsqueeze(2) * k2h).sum(-1)                         # [S,N,T,h]
    ta = torch.softmax(tl, dim=2)
    o = (ta.unsqueeze(-1) * v2h).sum(2).reshape(S, N, C)          # [S,N,C]
    out = _linear(o, w, "proj")
    if want_attn:
        # reference returns space_attn as [(S h), N, T, L]
        attn_maps = attn_maps.reshape(S * heads, N, T, L)
    return out, attn_maps


def trajectory_attention(query: Tensor, key: Tensor, value: Tensor, w: Weights, num_frames: int,
                         heads: int = 8, want_attn: bool = True, attn_keep: Optional[Tensor] = None
                         ) -> Tuple[Tensor, Optional[Tensor]]:
    """WC/temporal_attention.py:35-76 (TL/mmdet/models/plugins/msdeformattn_pixel_decoder.py:667-708).
    attn_keep [S, heads, N, T, L]: train-mode dropout factors (0 or 1/(1-p)) of the spatial attention map."""
    q = _linear(query, w, "q")
    k = _linear(key, w, "k")
    v = _linear(value, w, "v")
    return _trajectory_core(q, k, v, w, num_frames, heads, want_attn, attn_keep)


def cc_trajectory_attention(x: Tensor, w: Weights, seq_len: int, num_frames: int, heads: int = 8,
                            attn_keep: Optional[Tensor] = None) -> Tensor:
    """CC/maxtron_cross_clip_tracking_module.py:91-130 (fused qkv, no positional term).  attn_keep [B, heads, N, T, L]: train-mode
    dropout factors of the spatial attention map (:106)."""
    assert x.shape[1] == seq_len * num_frames
    C = x.shape[-1]
    qkv = _linear(x, w, "qkv")
    out, _ = _trajectory_core(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], w, num_frames, heads, False, attn_keep)
    return out


# --------------------------------------------------------------------------------------
# (a4) axial layer, (a5) encoder, (a8) Tube-Link gamma wrapper
# --------------------------------------------------------------------------------------
def axial_layer(src: Tensor, pos: Tensor, w: Weights, heads: int = 8, want_attn: bool = True, activation: str = "relu"
                ) -> Tuple[Tensor, Optional[Tensor], Optional[Tensor]]:
    """WC/temporal_attention.py:187-220.  src [(B T),(H W),C], pos [B,T,H,W,C].  activation: "relu" | "gelu" (:9-17, exact GELU)."""
    act = torch.relu if activation == "relu" else torch.nn.functional.gelu
    B, T, H, W, C = pos.shape
    x = src.reshape(B, T, H, W, C)
    # height pass: sequences (b, w), tokens (t, h)          (:197-204)
    xs = x.permute(0, 3, 1, 2, 4).reshape(B * W, T * H, C)
    ps = pos.permute(0, 3, 1, 2, 4).reshape(B * W, T * H, C).to(x.dtype)
    kq = xs + ps
    o, h_attn = trajectory_attention(kq, kq, xs, _sub(w, "height_attn"), T, heads, want_attn)
    xs = xs + o
    x = xs.reshape(B, W, T, H, C).permute(0, 2, 3, 1, 4)           # back to [B,T,H,W,C]
    # width pass: sequences (b, h), tokens (t, w)           (:206-213)
    xs = x.permute(0, 2, 1, 3, 4).reshape(B * H, T * W, C)
    ps = pos.permute(0, 2, 1, 3, 4).reshape(B * H, T * W, C).to(x.dtype)
    kq = xs + ps
    o, w_attn = trajectory_attention(kq, kq, xs, _sub(w, "width_attn"), T, heads, want_attn)
    xs = xs + o
    x = xs.reshape(B, H, T, W, C).permute(0, 2, 1, 3, 4).reshape(B * T, H * W, C)   # (:215)
    # norm1 + FFN + norm2                                   (:181-185, :217-218)
    x = _layer_norm(x, w, "norm1")
    ff = _linear(act(_linear(x, w, "linear1")), w, "linear2")
    x = _layer_norm(x + ff, w, "norm2")
    return x, h_attn, w_attn


def axial_pass(x: Tensor, pos: Tensor, w: Weights, which: int, heads: int = 8) -> Tensor:
    """One axial pass of the layer on a block [B,T,H,W,C] of the token grid (the two halves of `axial_layer`, for the off-axis
    sharding tests): which = 0 -> x + height_attn (WC/temporal_attention.py:197-204); 1 -> width_attn + norm1 + FFN + norm2 (:206-218)."""
    B, T, H, W, C = x.shape
    if which == 0:
        xs = x.permute(0, 3, 1, 2, 4).reshape(B * W, T * H, C)
        ps = pos.permute(0, 3, 1, 2, 4).reshape(B * W, T * H, C)
        y, _ = trajectory_attention(xs + ps, xs + ps, xs, _sub(w, "height_attn"), T, heads, want_attn=False)
        return (xs + y).reshape(B, W, T, H, C).permute(0, 2, 3, 1, 4).contiguous()
    xs = x.permute(0, 2, 1, 3, 4).reshape(B * H, T * W, C)
    ps = pos.permute(0, 2, 1, 3, 4).reshape(B * H, T * W, C)
    y, _ = trajectory_attention(xs + ps, xs + ps, xs, _sub(w, "width_attn"), T, heads, want_attn=False)
    z = _layer_norm((xs + y).reshape(B, H, T, W, C).permute(0, 2, 1, 3, 4), w, "norm1")
    return _layer_norm(z + _linear(torch.relu(_linear(z, w, "linear1")), w, "linear2"), w, "norm2").contiguous()


def dropout_keep(seed: int, site: int, count: int, p: float, dtype=torch.float64) -> Tensor:
    """The training tier's dropout factors (include/axvs.h): element `idx` of site `site` is kept iff
    (fmix32(fmix32(seed ^ site * 0x9E3779B9 ^ lo32(idx)) ^ hi32(idx)) >> 8) >= floor(p * 2^24); kept elements are scaled by
    1 / (1 - p).  Returns the flat factor vector [count] (0 or 1/(1-p)); p == 0 -> ones."""
    if p <= 0:
        return torch.ones(count, dtype=dtype)
    import numpy as np

    def fmix(h):
        h = h ^ (h >> np.uint32(16))
        h = h * np.uint32(0x85EBCA6B)
        h = h ^ (h >> np.uint32(13))
        h = h * np.uint32(0xC2B2AE35)
        return h ^ (h >> np.uint32(16))

    idx = np.arange(count, dtype=np.uint64)
    lo = (idx & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    hi = (idx >> np.uint64(32)).astype(np.uint32)
    h0 = np.uint32((int(seed) ^ ((int(site) * 0x9E3779B9) & 0xFFFFFFFF)) & 0xFFFFFFFF)
    h = fmix(fmix(h0 ^ lo) ^ hi)
    keep = (h >> np.uint32(8)) >= np.uint32(int(float(p) * 16777216.0))
    return torch.from_numpy(keep).to(dtype) / (1.0 - float(p))


def axial_layer_train(src: Tensor, pos: Tensor, w: Weights, heads: int, p_dropout: float, p_attn_drop: float, seed: int) -> Tensor:
    """TemporalAxialTrajectoryAttentionLayer.forward in train() mode (WC/temporal_attention.py:187-220) with the dropout factors
    of `dropout_keep` in place of torch's RNG: `dropout` (p_dropout) on the spatial attention maps (:55, :164-165) and in the
    FFN (dropout2 / dropout3, :182-183), `dropout1` (p_attn_drop) on both pass outputs (:204, :213).  Differentiable torch code:
    the gradient oracle of the training tier (tests call autograd on it)."""
    B, T, H, W, C = pos.shape
    dt = src.dtype
    x = src.reshape(B, T, H, W, C)

    def one_pass(xs, ps, name, site_attn, site_out):
        S, N, _ = xs.shape
        L = N // T
        keep = dropout_keep(seed, site_attn, S * heads * N * T * L, p_dropout, dt).reshape(S, heads, N, T, L) if p_dropout > 0 else None
        kq = xs + ps
        o, _ = trajectory_attention(kq, kq, xs, _sub(w, name), T, heads, want_attn=False, attn_keep=keep)
        return xs + o * dropout_keep(seed, site_out, S * N * C, p_attn_drop, dt).reshape(S, N, C)

    xs = one_pass(x.permute(0, 3, 1, 2, 4).reshape(B * W, T * H, C), pos.permute(0, 3, 1, 2, 4).reshape(B * W, T * H, C).to(dt),
                  "height_attn", 1, 2)
    x = xs.reshape(B, W, T, H, C).permute(0, 2, 3, 1, 4)
    xs = one_pass(x.permute(0, 2, 1, 3, 4).reshape(B * H, T * W, C), pos.permute(0, 2, 1, 3, 4).reshape(B * H, T * W, C).to(dt),
                  "width_attn", 3, 4)
    x = xs.reshape(B, H, T, W, C).permute(0, 2, 1, 3, 4).reshape(B * T, H * W, C)
    z = _layer_norm(x, w, "norm1")
    F_ = w["linear1.weight"].shape[0]
    M = B * T * H * W
    r = torch.relu(_linear(z, w, "linear1")) * dropout_keep(seed, 5, M * F_, p_dropout, dt).reshape(B * T, H * W, F_)
    ff = _linear(r, w, "linear2") * dropout_keep(seed, 6, M * C, p_dropout, dt).reshape(B * T, H * W, C)
    return _layer_norm(z + ff, w, "norm2")


def trajectory_layer(src: Tensor, pos: Tensor, w: Weights, heads: int = 8) -> Tensor:
    """TemporalTrajectoryAttentionLayer.forward, WC/temporal_attention.py:133-155: one trajectory attention over all T*H*W tokens
    of a clip (q = k = src + pos, v = src), residual, norm1, FFN, norm2.  src [(B T), HW, C], pos [B,T,H,W,C]."""
    B, T = pos.shape[:2]
    C = src.shape[-1]
    x = src.reshape(B, -1, C)
    kq = x + pos.reshape(B, -1, C)
    y, _ = trajectory_attention(kq, kq, x, _sub(w, "temporal_attn"), T, heads, want_attn=False)
    z = _layer_norm((x + y).reshape(src.shape), w, "norm1")
    return _layer_norm(z + _linear(torch.relu(_linear(z, w, "linear1")), w, "linear2"), w, "norm2")


def temporal_encoder(src: Tensor, pos: Tensor, w: Weights, num_layers: int, heads: int = 8,
                     want_attn: bool = True):
    """WC/temporal_attention.py:90-100: layers in sequence; attention maps of the last layer."""
    h_attn = w_attn = None
    for i in range(num_layers):
        src, h_attn, w_attn = axial_layer(src, pos, _sub(w, f"temporal_layers.{i}"), heads, want_attn)
    return src, h_attn, w_attn


def tubelink_temporal_residual(f: Tensor, pos3d: Tensor, gamma: Tensor, w: Weights, num_layers: int,
                               heads: int = 8) -> Tensor:
    """TL/mmdet/models/plugins/msdeformattn_pixel_decoder.py:623-627: f + gamma * encoder(f, pos)."""
    y, _, _ = temporal_encoder(f, pos3d, w, num_layers, heads, want_attn=False)
    return f + gamma.to(f.dtype) * y


# --------------------------------------------------------------------------------------
# (a6) 3-D sine positional embedding
# --------------------------------------------------------------------------------------
def pos_embed_sine_3d_masked(mask: Tensor, num_pos_feats: int, temperature: float = 10000.0, normalize: bool = True,
                             scale: float = 2 * math.pi, dtype: torch.dtype = torch.float32) -> Tensor:
    """WC/pos_embeddings.py:86-130 with a padding mask [B,T,H,W] (True = padded): the coordinates are running counts of the
    unmasked positions (:96-100), normalised by the last count of each axis (:101-105).  Channels-last [B,T,H,W,2*num_pos_feats]."""
    nm = (~mask.bool()).to(dtype)
    z, y, x = nm.cumsum(1), nm.cumsum(2), nm.cumsum(3)
    if normalize:
        eps = 1e-6
        z = z / (z[:, -1:] + eps) * scale
        y = y / (y[:, :, -1:] + eps) * scale
        x = x / (x[:, :, :, -1:] + eps) * scale
    n = num_pos_feats
    dim_t = torch.as_tensor(temperature, dtype=dtype) ** (2 * torch.floor(torch.arange(n, dtype=dtype) / 2) / n)
    dim_tz = torch.as_tensor(temperature, dtype=dtype) ** (2 * torch.floor(torch.arange(2 * n, dtype=dtype) / 2) / (2 * n))

    def interleave(coord: Tensor, dim: Tensor) -> Tensor:
        a = coord[..., None] / dim
        out = torch.empty_like(a)
        out[..., 0::2] = a[..., 0::2].sin()
        out[..., 1::2] = a[..., 1::2].cos()
        return out

    return torch.cat([interleave(y, dim_t), interleave(x, dim_t)], dim=-1) + interleave(z, dim_tz)


def pos_embed_sine_3d(B: int, T: int, H: int, W: int, num_pos_feats: int, temperature: float = 10000.0,
                      normalize: bool = True, scale: float = 2 * math.pi,
                      dtype: torch.dtype = torch.float32) -> Tensor:
    """WC/pos_embeddings.py:86-130 with mask=None.  Returns channels-last [B,T,H,W,2*num_pos_feats]."""
    z = torch.arange(1, T + 1, dtype=dtype)
    y = torch.arange(1, H + 1, dtype=dtype)
    xx = torch.arange(1, W + 1, dtype=dtype)
    if normalize:
        eps = 1e-6
        z = z / (z[-1] + eps) * scale
        y = y / (y[-1] + eps) * scale
        xx = xx / (xx[-1] + eps) * scale
    n = num_pos_feats
    i = torch.arange(n, dtype=dtype)
    dim_t = torch.as_tensor(temperature, dtype=dtype) ** (2 * torch.floor(i / 2) / n)
    iz = torch.arange(2 * n, dtype=dtype)
    dim_tz = torch.as_tensor(temperature, dtype=dtype) ** (2 * torch.floor(iz / 2) / (2 * n))

    def interleave(coord: Tensor, dim: Tensor) -> Tensor:
        a = coord[:, None] / dim                                   # [len, n]
        out = torch.empty_like(a)
        out[:, 0::2] = a[:, 0::2].sin()
        out[:, 1::2] = a[:, 1::2].cos()
        return out

    py = interleave(y, dim_t)       # [H, n]
    px = interleave(xx, dim_t)      # [W, n]
    pz = interleave(z, dim_tz)      # [T, 2n]
    pos = torch.empty(T, H, W, 2 * n, dtype=dtype)
    pos[..., :n] = py[None, :, None, :]
    pos[..., n:] = px[None, None, :, :]
    pos = pos + pz[:, None, None, :]
    return pos.unsqueeze(0).expand(B, T, H, W, 2 * n).contiguous()


# --------------------------------------------------------------------------------------
# cross-clip module (a9)-(a13)
# --------------------------------------------------------------------------------------
def _channels_first_ln(x: Tensor, weight: Tensor, bias: Tensor, eps: float = 1e-6) -> Tensor:
    """kmax_deeplab/modeling/backbone/convnext.py:73-81 for [B,C,L]."""
    u = x.mean(1, keepdim=True)
    s = ((x - u) ** 2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return weight.to(x.dtype)[:, None] * x + bias.to(x.dtype)[:, None]


def _batch_norm_eval(x: Tensor, w: Weights, name: str, eps: float = 1e-3) -> Tensor:
    """nn.SyncBatchNorm(eps=1e-3) in eval mode = affine with running stats
    (kmax_pixel_decoder.py:36-37).  Channel dim is 1."""
    shape = [1, -1] + [1] * (x.dim() - 2)
    rm = w[name + ".running_mean"].to(x.dtype).reshape(shape)
    rv = w[name + ".running_var"].to(x.dtype).reshape(shape)
    g = w[name + ".weight"].to(x.dtype).reshape(shape)
    b = w[name + ".bias"].to(x.dtype).reshape(shape)
    return (x - rm) / torch.sqrt(rv + eps) * g + b


def _conv1d_k1(x: Tensor, w: Weights, name: str) -> Tensor:
    """1x1 Conv1d on [B,Cin,L] -> [B,Cout,L]."""
    wt = w[name + ".weight"].to(x.dtype)[:, :, 0]
    y = torch.einsum("oc,bcl->bol", wt, x)
    b = w.get(name + ".bias")
    return y if b is None else y + b.to(x.dtype)[None, :, None]


def _gelu(x: Tensor) -> Tensor:
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def aspp(x: Tensor, w: Weights, kernel_sizes: Sequence[int], atrous_rates: Sequence[int], norm_fn: str = "ln") -> Tensor:
    """CC/...:176-201.  x [(B Q), C, Tc].  Dilated k-tap Conv1d over the clip axis with
    'same' replicate padding, concat, 1x1 proj (no bias) + norm + exact GELU."""
    Bq, C, Tc = x.shape
    branches = []
    for bi, (ks, r) in enumerate(zip(kernel_sizes, atrous_rates)):
        wt = w[f"_aspp_conv{bi}.weight"].to(x.dtype)               # [Cout, Cin, ks]
        bias = w[f"_aspp_conv{bi}.bias"].to(x.dtype)
        total = r * (ks - 1)
        left = total // 2                                          # torch 'same': left = total//2
        y = x.new_zeros(Bq, wt.shape[0], Tc)
        for tap in range(ks):
            idx = (torch.arange(Tc) - left + tap * r).clamp(0, Tc - 1)   # replicate padding
            y = y + torch.einsum("oc,bct->bot", wt[:, :, tap], x[:, :, idx])
        branches.append(y + bias[None, :, None])
    cat = torch.cat(branches, dim=1)
    y = _conv1d_k1(cat, w, "_proj_conv_bn_act.conv")
    if norm_fn == "ln":
        y = _channels_first_ln(y, w["_proj_conv_bn_act.norm.weight"], w["_proj_conv_bn_act.norm.bias"])
    elif norm_fn == "syncbn":
        y = _batch_norm_eval(y, w, "_proj_conv_bn_act.norm")
    else:                                                          # (the reference's ConvBN cannot be built with any other norm: kmax_pixel_decoder.py:68-69)
        raise ValueError(norm_fn)
    return _gelu(y)


def cc_layer(clip_query: Tensor, w: Weights, i: int, kernel_sizes, atrous_rates, norm_fn: str, heads: int = 8) -> Tensor:
    """One iteration of CC/...:286-297 on clip_query [B,Q,Tc,C] -> [B,Q,Tc,C]."""
    B, Q, Tc, C = clip_query.shape
    x = clip_query.permute(0, 2, 1, 3).reshape(B, Tc * Q, C)       # 'b q t c -> b (t q) c'
    wl = _sub(w, f"transformer_trajectory_self_attention_layers.{i}")
    x = _layer_norm(x + cc_trajectory_attention(x, _sub(wl, "self_attn"), Q, Tc, heads), wl, "norm")  # :156-161
    x = x.reshape(B, Tc, Q, C).permute(0, 2, 3, 1).reshape(B * Q, C, Tc)                              # '(b q) c t'
    y = aspp(x, _sub(w, f"conv_short_aggregate_layers.{i}"), kernel_sizes, atrous_rates, norm_fn) + x
    y = _layer_norm(y.transpose(1, 2), w, f"conv_norms.{i}")       # [(B Q), Tc, C]
    return y.reshape(B, Q, Tc, C)


def _conv_bn_gelu_1d(x: Tensor, w: Weights, name: str) -> Tensor:
    """ConvBN(256,256,k=1,bias=False,norm='syncbn',act='gelu',conv_type='1d') in eval (CC/...:266-270)."""
    return _gelu(_batch_norm_eval(_conv1d_k1(x, w, name + ".conv"), w, name + ".norm"))


def cc_predictor(mask_emb: Tensor, class_emb: Tensor, pixel_feature: Tensor, w: Weights,
                 num_clips: int, num_clip_frames: int) -> Tuple[Tensor, Tensor]:
    """CC/...:45-75 (eval branch).  mask_emb/class_emb [(B Tc), C, Q]; pixel_feature [(B Tc),128,(V H),W]."""
    act = torch.softmax(_conv1d_k1(class_emb, w, "_transformer_class_activation_head.conv"), dim=0)   # :48-49
    pooled = (class_emb * act).sum(0, keepdim=True)                                                    # :50
    logits = _conv1d_k1(pooled, w, "_transformer_class_head.conv").permute(0, 2, 1)                   # :51
    K1 = logits.shape[-1]
    void = logits.new_zeros(K1)
    void[-1] = math.log((K1 - 1) * 0.9 / (1 - 0.9))                # add_bias_towards_void, decoder :39-45
    logits = logits + void
    kern = _batch_norm_eval(_conv1d_k1(mask_emb, w, "_transformer_mask_head.conv"), w, "_transformer_mask_head.norm")
    masks = torch.einsum("bchw,bcn->bnhw", pixel_feature, kern)                                        # :61-67
    masks = _batch_norm_eval(masks.unsqueeze(1), w, "_pixel_space_mask_batch_norm").squeeze(1)         # :68
    BT, Qn, VH, Wd = masks.shape
    Bv = BT // num_clips
    H = VH // num_clip_frames
    masks = masks.reshape(Bv, num_clips, Qn, num_clip_frames, H, Wd).permute(0, 2, 1, 3, 4, 5)
    masks = masks.reshape(Bv, Qn, num_clips * num_clip_frames, H, Wd)                                  # :69
    return logits, masks


def cross_clip_module(clip_query: Tensor, panoptic_features: Tensor, w: Weights, num_layers: int,
                      num_clip_frames: int, kernel_sizes=(3, 3, 3), atrous_rates=(1, 2, 3), norm_fn: str = "ln",
                      heads: int = 8) -> Dict[str, object]:
    """CC/...:275-322 in eval mode (aux_outputs resized like :325-331)."""
    B, Q, Tc, C = clip_query.shape
    Bp, Cp, TV, H, W = panoptic_features.shape
    V = num_clip_frames
    pf = panoptic_features.reshape(Bp, Cp, Tc, V, H, W).permute(0, 2, 1, 3, 4, 5).reshape(Bp * Tc, Cp, V * H, W)
    cls_all, mask_all = [], []
    for i in range(num_layers):
        clip_query = cc_layer(clip_query, w, i, kernel_sizes, atrous_rates, norm_fn, heads)
        vq = clip_query.permute(0, 2, 3, 1).reshape(B * Tc, C, Q)                                     # '(b t) c q'
        ce = _conv_bn_gelu_1d(vq, w, "_class_embedding_projection")
        me = _conv_bn_gelu_1d(vq, w, "_mask_embedding_projection")
        lg, mk = cc_predictor(me, ce, pf, _sub(w, "_predictor"), Tc, V)
        cls_all.append(lg)
        mask_all.append(mk)
    size = mask_all[-1].shape[-3:]
    ac = size[-1] % 2 == 1
    aux = [{"pred_logits": a, "pred_masks": F.interpolate(b, size=size, mode="trilinear", align_corners=ac)}
           for a, b in zip(cls_all[:-1], mask_all[:-1])]
    return {"pred_logits": cls_all[-1], "pred_masks": mask_all[-1], "aux_outputs": aux, "clip_query": clip_query}


# ---- the cross-clip module in train() mode: differentiable torch code, the gradient oracle of the training tier ----
def cc_module_param_shapes(num_layers: int, num_classes: int) -> Dict[str, Tuple[int, ...]]:
    """state_dict keys and shapes of CrossClipTrackingModule(num_layers, num_classes, norm_fn='ln', kernel sizes 3) (CC:233-272)."""
    C, Cm, K1 = 256, 128, num_classes + 1
    sh: Dict[str, Tuple[int, ...]] = {}
    for i in range(num_layers):
        p = f"transformer_trajectory_self_attention_layers.{i}."
        for name, out in (("qkv", 3 * C), ("proj_q", C), ("proj_kv", 2 * C), ("proj", C)):
            sh[p + f"self_attn.{name}.weight"] = (out, C)
            sh[p + f"self_attn.{name}.bias"] = (out,)
        sh[p + "norm.weight"] = (C,)
        sh[p + "norm.bias"] = (C,)
    for i in range(num_layers):
        p = f"conv_short_aggregate_layers.{i}."
        for k in range(3):
            sh[p + f"_aspp_conv{k}.weight"] = (C, C, 3)
            sh[p + f"_aspp_conv{k}.bias"] = (C,)
        sh[p + "_proj_conv_bn_act.conv.weight"] = (C, 3 * C, 1)
        sh[p + "_proj_conv_bn_act.norm.weight"] = (C,)
        sh[p + "_proj_conv_bn_act.norm.bias"] = (C,)
    for i in range(num_layers):
        sh[f"conv_norms.{i}.weight"] = (C,)
        sh[f"conv_norms.{i}.bias"] = (C,)

    def bn(prefix, n):
        for k in ("weight", "bias", "running_mean", "running_var"):
            sh[f"{prefix}.{k}"] = (n,)

    for name in ("_class_embedding_projection", "_mask_embedding_projection"):
        sh[name + ".conv.weight"] = (C, C, 1)
        bn(name + ".norm", C)
    sh["_predictor._transformer_mask_head.conv.weight"] = (Cm, C, 1)
    bn("_predictor._transformer_mask_head.norm", Cm)
    sh["_predictor._transformer_class_head.conv.weight"] = (K1, C, 1)
    sh["_predictor._transformer_class_head.conv.bias"] = (K1,)
    sh["_predictor._transformer_class_activation_head.conv.weight"] = (1, C, 1)
    sh["_predictor._transformer_class_activation_head.conv.bias"] = (1,)
    bn("_predictor._pixel_space_mask_batch_norm", 1)
    return sh


def cc_module_train(clip_query, panoptic_features, w: Weights, num_layers: int, num_clip_frames: int,
                    atrous_rates=(1, 2, 3), p_attn_drop: float = 0.0, p_aspp_drop: float = 0.0, seed: int = 0, heads: int = 8):
    """CrossClipTrackingModule.forward in train() mode (CC:275-322 with the predictor's training branch :53-57), dropout factors
    from `dropout_keep` (site 10 + 2 l: attention maps of layer l, CC:106; 11 + 2 l: the ASPP's _proj_drop, CC:199).
    -> (class logits per layer [1,Q,K1], mask logits per layer [B,Q,Tc*V,H,W], {bn site: [(batch mean, unbiased var) per layer]}).
    With LISTS of clip_query / panoptic_features (one entry per data-parallel rank) the BatchNorm statistics run over all ranks'
    rows, as nn.SyncBatchNorm computes them, and the first two results are lists over ranks."""
    ranks = isinstance(clip_query, (list, tuple))
    cqs = list(clip_query) if ranks else [clip_query]
    pfs_in = list(panoptic_features) if ranks else [panoptic_features]
    B, Q, Tc, C = cqs[0].shape
    Bp, Cp, TV, H, W = pfs_in[0].shape
    V = num_clip_frames
    dt = cqs[0].dtype
    pfs = [pf.reshape(Bp, Cp, Tc, V, H, W).permute(0, 2, 1, 3, 4, 5).reshape(Bp * Tc, Cp, V * H, W) for pf in pfs_in]          # :278
    cls_all = [[] for _ in cqs]
    mask_all = [[] for _ in cqs]
    stats: Dict[str, list] = {}

    def bn(xs, name, eps=1e-3):
        """(Sync)BatchNorm in train mode over every rank's rows: batch mean, biased variance (kmax_pixel_decoder.py:36-37)."""
        Cn = xs[0].shape[1]
        shape = [1, -1] + [1] * (xs[0].dim() - 2)
        flat = torch.cat([x.transpose(0, 1).reshape(Cn, -1) for x in xs], dim=1)
        n = flat.shape[1]
        mean = flat.mean(1)
        var = ((flat - mean[:, None]) ** 2).mean(1)
        stats.setdefault(name, []).append((mean.detach(), (var * n / max(n - 1, 1)).detach()))
        g = w[name + ".weight"].to(dt).reshape(shape)
        b = w[name + ".bias"].to(dt).reshape(shape)
        return [(x - mean.reshape(shape)) / torch.sqrt(var.reshape(shape) + eps) * g + b for x in xs]

    wp = _sub(w, "_predictor")
    for i in range(num_layers):
        wl = _sub(w, f"transformer_trajectory_self_attention_layers.{i}")
        N = Tc * Q
        keep = dropout_keep(seed, 10 + 2 * i, B * heads * N * Tc * Q, p_attn_drop, dt).reshape(B, heads, N, Tc, Q) if p_attn_drop > 0 else None
        nxt = []
        for cq in cqs:
            x = cq.permute(0, 2, 1, 3).reshape(B, Tc * Q, C)                                                               # :284
            x = _layer_norm(x + cc_trajectory_attention(x, _sub(wl, "self_attn"), Q, Tc, heads, keep), wl, "norm")           # :156-161
            x = x.reshape(B, Tc, Q, C).permute(0, 2, 3, 1).reshape(B * Q, C, Tc)                                           # :290
            a = aspp(x, _sub(w, f"conv_short_aggregate_layers.{i}"), (3, 3, 3), atrous_rates, "ln")
            a = a * dropout_keep(seed, 11 + 2 * i, a.numel(), p_aspp_drop, dt).reshape(a.shape)                             # :199
            nxt.append(_layer_norm((a + x).transpose(1, 2), w, f"conv_norms.{i}").reshape(B, Q, Tc, C))                    # :293-297
        cqs = nxt
        vqs = [cq.permute(0, 2, 3, 1).reshape(B * Tc, C, Q) for cq in cqs]                                                  # :298
        ces = [_gelu(y) for y in bn([_conv1d_k1(vq, w, "_class_embedding_projection.conv") for vq in vqs], "_class_embedding_projection.norm")]
        mes = [_gelu(y) for y in bn([_conv1d_k1(vq, w, "_mask_embedding_projection.conv") for vq in vqs], "_mask_embedding_projection.norm")]
        kerns = bn([_conv1d_k1(me, wp, "_transformer_mask_head.conv") for me in mes], "_predictor._transformer_mask_head.norm")   # :53
        pre = [torch.einsum("bchw,bcn->bnhw", pf, kern).unsqueeze(1) for pf, kern in zip(pfs, kerns)]                       # :55
        post = bn(pre, "_predictor._pixel_space_mask_batch_norm")                                                          # :56
        for r, (ce, masks) in enumerate(zip(ces, post)):
            act = torch.softmax(_conv1d_k1(ce, wp, "_transformer_class_activation_head.conv"), dim=0)                      # :48-49
            pooled = (ce * act).sum(0, keepdim=True)                                                                       # :50
            logits = _conv1d_k1(pooled, wp, "_transformer_class_head.conv").permute(0, 2, 1)                               # :51
            K1 = logits.shape[-1]
            void = logits.new_zeros(K1)
            void[-1] = math.log((K1 - 1) * 0.9 / (1 - 0.9))
            cls_all[r].append(logits + void)                                                                               # :52
            mask_all[r].append(masks.squeeze(1).reshape(B, Tc, Q, V, H, W).permute(0, 2, 1, 3, 4, 5).reshape(B, Q, Tc * V, H, W))   # :57
    if ranks:
        return cls_all, mask_all, stats
    return cls_all[0], mask_all[0], stats


# --------------------------------------------------------------------------------------
# Tube-Link flavour of the cross-clip module: the same trajectory-attention / ASPP core, different heads.
# TLCC = MaXTron_Tube-Link/models/video/tube_link_vis/mask2former_video_cc_head.py
# --------------------------------------------------------------------------------------
def tl_pred_class(x: Tensor, w: Weights) -> Tensor:
    """TLCC:783-797 for one decoder layer.  x [B,Tc,Q,C] (already post-normed) -> class logits [B,Q,K+1]:
    Linear(C,1) activation, softmax over the clips, weighted sum of the queries over clips, Linear(C,K+1)."""
    act = torch.softmax(_linear(x, w, "activation_proj"), dim=1)     # [B,Tc,Q,1], softmax over clips
    pooled = (x * act).sum(dim=1)                                      # [B,Q,C]
    return _linear(pooled, w, "cls_embed")


def tl_forward_head_clips(x: Tensor, mask_feature: Tensor, w: Weights) -> Tuple[Tensor, Tensor]:
    """TLCC:761-781 for one decoder layer.  x [B,Q,Tc,C] (output of a cross-clip layer), mask_feature [B,T,Cm,h,w] with
    T = Tc * frames_per_clip  ->  (class logits [B,Q,K+1], mask logits [B,T,Q,h,w])."""
    B, Q, Tc, C = x.shape
    T = mask_feature.shape[1]
    fpc = T // Tc
    xn = _layer_norm(x, w, "transformer_decoder.post_norm").permute(0, 2, 1, 3)        # [B,Tc,Q,C]   :768-769
    cls = tl_pred_class(xn, w)
    me = _linear(torch.relu(_linear(torch.relu(_linear(xn, w, "mask_embed.0")), w, "mask_embed.2")), w, "mask_embed.4")
    masks = []
    for c in range(Tc):                                                                 # :774-778
        mf = mask_feature[:, fpc * c:fpc * (c + 1)]                                     # [B,fpc,Cm,h,w]
        masks.append(torch.einsum("bqc,btchw->btqhw", me[:, c], mf))
    return cls, torch.cat(masks, dim=1)


def tl_cross_clip_head(clip_query: Tensor, mask_features: Tensor, w: Weights, num_layers: int, kernel_sizes=(3, 3, 3),
                       atrous_rates=(1, 2, 3), norm_fn: str = "ln", heads: int = 8):
    """TLCC:925-950: clip_query [B,Tc,Q,C] (matched clip queries), mask_features [B,T,Cm,h,w] ->
    (list of class logits [B,Q,K+1] per layer, list of mask logits [B,T,Q,h,w] per layer)."""
    x = clip_query.permute(0, 2, 1, 3)                                 # carried here as [B,Q,Tc,C] (same tokens as 'b c t q')
    cls_all, mask_all = [], []
    for i in range(num_layers):
        x = cc_layer(x, w, i, kernel_sizes, atrous_rates, norm_fn, heads)
        c, m = tl_forward_head_clips(x, mask_features, w)
        cls_all.append(c)
        mask_all.append(m)
    return cls_all, mask_all


# --------------------------------------------------------------------------------------
# Multi-scale deformable attention.  OPS = MaXTron_Video-kMaX/maxtron_deeplab/modeling/
# within_clip_tracking_module/ops
# --------------------------------------------------------------------------------------
def msda_core(value: Tensor, spatial_shapes, sampling_locations: Tensor, attention_weights: Tensor) -> Tensor:
    """OPS/functions/ms_deform_attn_func.py:55-77 (and the CUDA op OPS/src/cuda/ms_deform_attn_im2col_cuda.cuh, which
    samples at (loc*size - 0.5) with zero padding) restated with explicit bilinear gathers -- no grid_sample.
    value [N,S,M,D]; spatial_shapes [(H,W)]*L; sampling_locations [N,Lq,M,L,P,2] (x,y in [0,1]); attention_weights
    [N,Lq,M,L,P] -> [N,Lq,M*D]."""
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_locations.shape
    out = value.new_zeros(N, Lq, M, D)
    start = 0
    n_idx = torch.arange(N).view(N, 1, 1, 1).expand(N, Lq, M, P)
    m_idx = torch.arange(M).view(1, 1, M, 1).expand(N, Lq, M, P)
    for lvl, (H, W) in enumerate(spatial_shapes):
        H, W = int(H), int(W)
        loc = sampling_locations[:, :, :, lvl]                        # [N,Lq,M,P,2]
        x = loc[..., 0] * W - 0.5
        y = loc[..., 1] * H - 0.5
        x0, y0 = torch.floor(x), torch.floor(y)
        fx, fy = x - x0, y - y0
        aw = attention_weights[:, :, :, lvl]                          # [N,Lq,M,P]
        for dy, wy in ((0, 1 - fy), (1, fy)):
            for dx, wx in ((0, 1 - fx), (1, fx)):
                xi, yi = (x0 + dx).long(), (y0 + dy).long()
                ok = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)
                idx = start + yi.clamp(0, H - 1) * W + xi.clamp(0, W - 1)
                v = value[n_idx, idx, m_idx]                          # [N,Lq,M,P,D]
                out = out + (v * (wy * wx * aw * ok.to(value.dtype)).unsqueeze(-1)).sum(3)
        start += H * W
    return out.reshape(N, Lq, M * D)


def msda_module(query: Tensor, reference_points: Tensor, input_flatten: Tensor, spatial_shapes, w: Weights, n_heads: int,
                n_levels: int, n_points: int, input_padding_mask: Optional[Tensor] = None) -> Tensor:
    """MSDeformAttn.forward, OPS/modules/ms_deform_attn.py:81-125 (level_start_index follows from spatial_shapes)."""
    N, Lq, C = query.shape
    S = input_flatten.shape[1]
    value = _linear(input_flatten, w, "value_proj")
    if input_padding_mask is not None:
        value = value.masked_fill(input_padding_mask[..., None], 0.0)
    value = value.reshape(N, S, n_heads, C // n_heads)
    off = _linear(query, w, "sampling_offsets").reshape(N, Lq, n_heads, n_levels, n_points, 2)
    aw = torch.softmax(_linear(query, w, "attention_weights").reshape(N, Lq, n_heads, n_levels * n_points), -1)
    aw = aw.reshape(N, Lq, n_heads, n_levels, n_points)
    shp = torch.as_tensor([[int(h), int(ww)] for h, ww in spatial_shapes], dtype=query.dtype)
    if reference_points.shape[-1] == 2:
        normalizer = torch.stack([shp[:, 1], shp[:, 0]], -1)          # (W, H) per level
        loc = reference_points[:, :, None, :, None, :] + off / normalizer[None, None, None, :, None, :]
    elif reference_points.shape[-1] == 4:
        loc = reference_points[:, :, None, :, None, :2] + off / n_points * reference_points[:, :, None, :, None, 2:] * 0.5
    else:
        raise ValueError("Last dim of reference_points must be 2 or 4")
    return _linear(msda_core(value, spatial_shapes, loc, aw), w, "output_proj")


def tl_plugin_attention(query: Tensor, query_pos: Optional[Tensor], pos3d, reference_points: Tensor, spatial_shapes, w: Weights,
                        n_heads: int, n_points: int, num_temporal_levels: int, num_temporal_layers: int, skip_connect: bool = True,
                        key_padding_mask: Optional[Tensor] = None) -> Tensor:
    """MultiScaleDeformableAxialTrajectoryAttention.forward (TL/mmdet/models/plugins/msdeformattn_pixel_decoder.py:566-638), batch-first
    tensors [bs, num_query, C], value = identity = query, eval (dropout = identity):
    value_proj / offsets / softmaxed weights / bilinear sampling (:589-611) -> per temporal level `f + gamma * encoder(f, pos3d[i])`
    (:613-630) -> output_proj + identity (:633-638)."""
    N, Lq, C = query.shape
    n_levels = len(spatial_shapes)
    q = query if query_pos is None else query + query_pos
    value = _linear(query, w, "value_proj")
    if key_padding_mask is not None:
        value = value.masked_fill(key_padding_mask[..., None], 0.0)
    value = value.reshape(N, Lq, n_heads, C // n_heads)
    off = _linear(q, w, "sampling_offsets").reshape(N, Lq, n_heads, n_levels, n_points, 2)
    aw = torch.softmax(_linear(q, w, "attention_weights").reshape(N, Lq, n_heads, n_levels * n_points), -1)
    aw = aw.reshape(N, Lq, n_heads, n_levels, n_points)
    shp = torch.as_tensor([[int(h), int(ww)] for h, ww in spatial_shapes], dtype=query.dtype)
    if reference_points.shape[-1] == 2:
        normalizer = torch.stack([shp[:, 1], shp[:, 0]], -1)
        loc = reference_points[:, :, None, :, None, :] + off / normalizer[None, None, None, :, None, :]
    else:
        loc = reference_points[:, :, None, :, None, :2] + off / n_points * reference_points[:, :, None, :, None, 2:] * 0.5
    sampled = msda_core(value, spatial_shapes, loc, aw)
    outs, start = [], 0
    for i, (h, ww) in enumerate(spatial_shapes):
        f = sampled[:, start:start + h * ww]
        start += h * ww
        if i < num_temporal_levels:
            y, _, _ = temporal_encoder(f, pos3d[i], _sub(w, "temporal_layer"), num_temporal_layers, n_heads, want_attn=False)
            f = f + w["gamma"].to(f.dtype) * y if skip_connect else y
        outs.append(f)
    return _linear(torch.cat(outs, 1), w, "output_proj") + query


def msda_encoder_layer(src: Tensor, pos: Optional[Tensor], reference_points: Tensor, spatial_shapes, w: Weights, n_heads: int,
                       n_levels: int, n_points: int, padding_mask: Optional[Tensor] = None) -> Tensor:
    """MSDeformAttnTransformerEncoderLayer.forward, WC/msdeformattn.py:207-216 (eval: dropouts are identities)."""
    q = src if pos is None else src + pos
    x = src + msda_module(q, reference_points, src, spatial_shapes, _sub(w, "self_attn"), n_heads, n_levels, n_points, padding_mask)
    x = _layer_norm(x, w, "norm1")
    return _layer_norm(x + _linear(torch.relu(_linear(x, w, "linear1")), w, "linear2"), w, "norm2")


# --------------------------------------------------------------------------------------
# Within-clip pixel decoder around the stages: WC/msdeformattn.py:34-174 (encoder-only transformer),
# :219-275 (stage loop), :293-437 (MSDeformAttnPixelDecoder), WC/pos_embeddings.py:12-53 (2-D sine embedding)
# --------------------------------------------------------------------------------------
def pos_embed_sine_2d(N: int, H: int, W: int, num_pos_feats: int, temperature: float = 10000.0, normalize: bool = True,
                      scale: float = 2 * math.pi, dtype=torch.float32) -> Tensor:
    """WC/pos_embeddings.py:29-53 with mask=None -> channels-last [N,H,W,2*num_pos_feats] (y half, then x half)."""
    ye = torch.arange(1, H + 1, dtype=dtype).view(H, 1).expand(H, W)
    xe = torch.arange(1, W + 1, dtype=dtype).view(1, W).expand(H, W)
    if normalize:
        eps = 1e-6
        ye = ye / (H + eps) * scale
        xe = xe / (W + eps) * scale
    i = torch.arange(num_pos_feats, dtype=dtype)
    dim_t = temperature ** (2 * torch.div(i, 2, rounding_mode="trunc") / num_pos_feats)

    def enc(e):
        a = e[..., None] / dim_t
        return torch.stack((a[..., 0::2].sin(), a[..., 1::2].cos()), dim=-1).flatten(-2)

    return torch.cat((enc(ye), enc(xe)), dim=-1)[None].expand(N, H, W, 2 * num_pos_feats)


def conv1x1_group_norm(x: Tensor, w: Weights, name: str, groups: int = 32, eps: float = 1e-5) -> Tensor:
    """nn.Sequential(Conv2d(k=1), GroupNorm(32, C)) (WC/msdeformattn.py:355-362) on NCHW x."""
    wt, b = w[name + ".0.weight"], w[name + ".0.bias"]
    y = torch.einsum("nchw,oc->nohw", x, wt.reshape(wt.shape[0], wt.shape[1])) + b.view(1, -1, 1, 1)
    N, C, H, W = y.shape
    yg = y.reshape(N, groups, C // groups * H * W)
    mu = yg.mean(-1, keepdim=True)
    var = ((yg - mu) ** 2).mean(-1, keepdim=True)
    yn = ((yg - mu) / torch.sqrt(var + eps)).reshape(N, C, H, W)
    return yn * w[name + ".1.weight"].view(1, -1, 1, 1) + w[name + ".1.bias"].view(1, -1, 1, 1)


def pixel_decoder(features: Dict[str, Tensor], w: Weights, spatial_in: Sequence[str], temporal_in: Sequence[str], num_stages: int,
                  temporal_layers_per_stage: int, heads: int = 8, n_points: int = 4, num_clip_frames: int = 1,
                  B: int = 1, trace: Optional[list] = None, with_spatial: bool = True) -> Dict[str, Tensor]:
    """MSDeformAttnPixelDecoder.forward_features (WC/msdeformattn.py:404-437) with spatial and temporal layers in every stage.
    `trace` (a list) receives ("setup", {pos, pos3d, ref, shapes, order}, src) and then (tag, input, output) of every stage's
    spatial layer and temporal encoders -- the teacher inputs of the per-stage parity tests.
    `with_spatial` False: the temporal-only decoder (SPATIAL_LAYERS 0: TemporalTransformerEncoder, WC/msdeformattn.py:276-290).
    `spatial_in` / `temporal_in`: feature names sorted by stride (high resolution first), as the constructor sorts them (:344-349).
    features[name]: [(B T), C_l, H_l, W_l].  Returns {name: [(B T), C_l, H_l, W_l]}."""
    order = list(spatial_in)[::-1]                                    # low -> high resolution (:411)
    L, Lt = len(order), len(temporal_in)
    BT = features[order[0]].shape[0]
    T = BT // B
    C = w["input_proj.0.0.weight"].shape[0]
    srcs, poss, pos3d, shapes = [], [], [], []
    for idx, f in enumerate(order):
        x = features[f]
        y = conv1x1_group_norm(x, w, f"input_proj.{idx}")
        _, _, H, W = y.shape
        shapes.append((H, W))
        srcs.append(y.flatten(2).transpose(1, 2))                     # [(B T), HW, C]
        if with_spatial:
            poss.append(pos_embed_sine_2d(BT, H, W, C // 2, dtype=x.dtype).reshape(BT, H * W, C)
                        + w["transformer.level_embed_2d"][idx].view(1, 1, -1))
        if f in temporal_in:
            pos3d.append(pos_embed_sine_3d(B, T, H, W, C // 2, dtype=x.dtype) + w["transformer.level_embed_3d"][len(pos3d)].view(1, 1, 1, 1, -1))
    src = torch.cat(srcs, 1)
    pos = torch.cat(poss, 1) if with_spatial else None
    refs = []
    for (H, W) in shapes:                                             # get_reference_points, valid ratios = 1 (:229-242)
        ys = (torch.arange(H, dtype=src.dtype) + 0.5) / H
        xs = (torch.arange(W, dtype=src.dtype) + 0.5) / W
        refs.append(torch.stack((xs.view(1, W).expand(H, W).reshape(-1), ys.view(H, 1).expand(H, W).reshape(-1)), -1))
    ref = torch.cat(refs, 0)[None, :, None, :].expand(BT, -1, L, 2)
    sizes = [h * ww for h, ww in shapes]
    out = src
    if trace is not None:
        trace.append(("setup", dict(pos=pos, pos3d=pos3d, ref=ref, shapes=shapes, order=order), src))
    for s_i in range(num_stages):                                     # :247-266
        if with_spatial:
            x_in = out
            out = msda_encoder_layer(out, pos, ref, shapes, _sub(w, f"transformer.encoder.spatial_layers.{s_i}"), heads, L, n_points)
            if trace is not None:
                trace.append((f"s{s_i}_spatial", x_in, out))
        parts = list(torch.split(out, sizes, dim=1))
        for i in range(Lt):
            x_in = parts[i]
            parts[i], _, _ = temporal_encoder(parts[i], pos3d[i], _sub(w, f"transformer.encoder.temporal_layers.{s_i}"),
                                              temporal_layers_per_stage, heads, want_attn=False)
            if trace is not None:
                trace.append((f"s{s_i}_temporal_{order[i]}", x_in, parts[i]))
        out = torch.cat(parts, 1)
    res = {}
    for i, z in enumerate(torch.split(out, sizes, dim=1)):            # :426-435
        H, W = shapes[i]
        zz = z.transpose(1, 2).reshape(BT, C, H, W)
        res[order[i]] = conv1x1_group_norm(zz, w, f"output_proj.{i}")
    return res


# --------------------------------------------------------------------------------------
# synthetic inputs / weights shared by tests, smoke and bench (SURVEY.md 8d recipe)
# --------------------------------------------------------------------------------------
def axial_layer_param_shapes(C: int, d_ffn: int) -> Dict[str, Tuple[int, ...]]:
    shapes: Dict[str, Tuple[int, ...]] = {}
    for ax in ("height_attn", "width_attn"):
        for name, out in (("q", C), ("k", C), ("v", C), ("proj_q", C), ("proj_kv", 2 * C), ("proj", C)):
            shapes[f"{ax}.{name}.weight"] = (out, C)
            shapes[f"{ax}.{name}.bias"] = (out,)
    shapes["norm1.weight"] = (C,)
    shapes["norm1.bias"] = (C,)
    shapes["linear1.weight"] = (d_ffn, C)
    shapes["linear1.bias"] = (d_ffn,)
    shapes["linear2.weight"] = (C, d_ffn)
    shapes["linear2.bias"] = (C,)
    shapes["norm2.weight"] = (C,)
    shapes["norm2.bias"] = (C,)
    return shapes


def random_weights(shapes: Dict[str, Tuple[int, ...]], seed: int) -> Weights:
    """xavier_uniform for matrices, U(-0.1,0.1) for biases, LN gains ~ 1 + U(-0.1,0.1).
    Deterministic in (shapes order, seed); independent of torch's global RNG."""
    g = torch.Generator().manual_seed(seed)
    out: Weights = {}
    for name, shp in shapes.items():
        if len(shp) >= 2:
            fan_out = shp[0] * (int(torch.tensor(shp[2:]).prod()) if len(shp) > 2 else 1)
            fan_in = shp[1] * (int(torch.tensor(shp[2:]).prod()) if len(shp) > 2 else 1)
            bound = math.sqrt(6.0 / (fan_in + fan_out))
            out[name] = (torch.rand(shp, generator=g) * 2 - 1) * bound
        elif name.endswith("running_var"):
            out[name] = torch.rand(shp, generator=g) * 0.5 + 0.75
        elif "norm" in name and name.endswith("weight"):
            out[name] = 1.0 + (torch.rand(shp, generator=g) * 2 - 1) * 0.1
        else:
            out[name] = (torch.rand(shp, generator=g) * 2 - 1) * 0.1
    return out


def synthetic_clip(B: int, T: int, C: int, H: int, W: int, seed: int = 0) -> Tuple[Tensor, Tensor]:
    """x ~ N(0,1) [B,T,C,H,W] -> src [(B T),(H W),C]; pos = 3-D sine embedding [B,T,H,W,C]."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, T, C, H, W, generator=g)
    src = x.permute(0, 1, 3, 4, 2).reshape(B * T, H * W, C).contiguous()
    pos = pos_embed_sine_3d(B, T, H, W, C // 2)
    return src, pos
