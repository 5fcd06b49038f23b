"""Generate tests/golden/*.npz by running the REFERENCE modules (imported from /root/reference).

Runs only in the build container (the reference never travels to the GPU box).  Weights and
inputs are produced by ``axvs_oracle.random_weights`` / torch generators from fixed seeds and
loaded into the reference modules with ``load_state_dict(strict=True)``, so a fixture only
has to carry: the seeds, the parameter name->shape table, a weight checksum (to detect RNG
drift) and the reference outputs (full for small shapes, strided subsample + float64
checksums for the BASELINE-sized ones).

    python oracle/gen_golden.py            # rewrites tests/golden/*.npz
"""
from __future__ import annotations

import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import axvs_oracle as orc  # noqa: E402

VK = "/root/reference/MaXTron_Video-kMaX"
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


# ---- reference loader (no package __init__ is executed; only decorators / registries / init helpers are stubbed)
def _mod(name, **kw):
    m = types.ModuleType(name)
    m.__dict__.update(kw)
    sys.modules[name] = m
    return m


def _load(dotted):
    path = f"{VK}/{dotted.replace('.', '/')}.py"
    spec = importlib.util.spec_from_file_location(dotted, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[dotted] = m
    spec.loader.exec_module(m)
    return m


def load_reference():
    WC = "maxtron_deeplab.modeling.within_clip_tracking_module"
    for p in ["maxtron_deeplab", "maxtron_deeplab.modeling", WC]:
        _mod(p).__path__ = [f"{VK}/{p.replace('.', '/')}"]
    ta = _load(WC + ".temporal_attention")
    pe = _load(WC + ".pos_embeddings")

    class _Reg:
        def __init__(s, *a):
            pass

        def register(s, *a, **k):
            return (lambda c: c) if not a else a[0]

    class ShapeSpec:
        def __init__(s, channels=None, stride=None):
            s.channels, s.stride = channels, stride

    class DropPath(torch.nn.Module):
        def __init__(s, p=0.0):
            super().__init__()

        def forward(s, x):
            return x

    _mod("timm"); _mod("timm.models")
    _mod("timm.models.layers", DropPath=DropPath, trunc_normal_tf_=torch.nn.init.trunc_normal_)
    _mod("detectron2").__path__ = []
    _mod("detectron2.config", configurable=lambda f=None, **k: f)
    _mod("detectron2.layers", ShapeSpec=ShapeSpec)
    _mod("detectron2.utils").__path__ = []
    _mod("detectron2.utils.registry", Registry=_Reg)
    _mod("detectron2.modeling", SEM_SEG_HEADS_REGISTRY=_Reg(), BACKBONE_REGISTRY=_Reg(), Backbone=torch.nn.Module,
         ShapeSpec=ShapeSpec)
    for p in ["kmax_deeplab", "kmax_deeplab.modeling", "kmax_deeplab.modeling.pixel_decoder",
              "kmax_deeplab.modeling.backbone", "kmax_deeplab.modeling.transformer_decoder",
              "maxtron_deeplab.modeling.transformer_decoder", "maxtron_deeplab.modeling.cross_clip_tracking_module"]:
        _mod(p).__path__ = [f"{VK}/{p.replace('.', '/')}"]
    cc = _load("maxtron_deeplab.modeling.cross_clip_tracking_module.maxtron_cross_clip_tracking_module")
    return ta, pe, cc


# ---- helpers
def shapes_of(module):
    return {k: tuple(v.shape) for k, v in module.state_dict().items() if v.dtype.is_floating_point}


def load_random(module, seed):
    shapes = shapes_of(module)
    w = orc.random_weights(shapes, seed)
    sd = module.state_dict()
    for k, v in w.items():
        sd[k] = v
    module.load_state_dict(sd, strict=True)
    return shapes, w


def wsum(w):
    return float(sum(v.double().sum() for v in w.values()))


def meta(**kw):
    return np.frombuffer(json.dumps(kw).encode(), dtype=np.uint8)


def checks(t):
    t = t.double()
    return np.array([t.sum().item(), (t * t).sum().item(), t.abs().max().item()], dtype=np.float64)


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().numpy() if isinstance(v, torch.Tensor) else v) for k, v in arrs.items()})
    print(f"{name}: {os.path.getsize(path) / 1e3:.1f} kB")


def main():
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    ta, pe, cc = load_reference()

    # G1: TrajectoryAttention (q/k/v flavour)
    for (S, T, L, C) in ([] if "--traj-layer-only" in sys.argv else [(3, 2, 7, 64), (2, 5, 6, 64), (2, 1, 9, 64), (4, 4, 16, 256)]):
        m = ta.TrajectoryAttention(C, num_heads=8).eval()
        seed = 1000 + S * 100 + T * 10 + L
        shapes, w = load_random(m, seed)
        g = torch.Generator().manual_seed(seed + 1)
        query = torch.randn(S, T * L, C, generator=g)
        value = torch.randn(S, T * L, C, generator=g)
        out, attn = m(query, query, value, num_frames=T)
        save(f"g1_traj_S{S}_T{T}_L{L}_C{C}", meta=meta(S=S, T=T, L=L, C=C, heads=8, seed=seed, shapes=shapes),
             wsum=np.float64(wsum(w)), out=out, attn=attn if attn.numel() < 300000 else attn[::4],
             attn_stride=np.int64(1 if attn.numel() < 300000 else 4), attn_checks=checks(attn))

    # G2/G3: axial layer + positional embedding; G4: encoder
    for (B, T, C, H, W, dffn, full) in ([] if "--traj-layer-only" in sys.argv else
                                        [(1, 2, 128, 32, 32, 1024, True), (2, 3, 64, 5, 7, 256, True),
                                         (1, 5, 64, 6, 4, 256, True), (1, 4, 256, 64, 64, 1024, False),
                                         (1, 1, 64, 4, 5, 128, True)]):
        layer = ta.TemporalAxialTrajectoryAttentionLayer(d_model=C, d_ffn=dffn, n_heads=8).eval()
        seed = 2000 + B * 1000 + T * 100 + H
        shapes, w = load_random(layer, seed)
        g = torch.Generator().manual_seed(seed + 1)
        x = torch.randn(B, T, C, H, W, generator=g)
        src = x.permute(0, 1, 3, 4, 2).reshape(B * T, H * W, C).contiguous()
        pos = pe.PositionEmbeddingSine3D(C // 2, normalize=True)(x, fmt="btchw").permute(0, 1, 3, 4, 2).contiguous()
        out, ha, wa = layer(src, pos)
        stride = 1 if full else 16
        arrs = dict(meta=meta(B=B, T=T, C=C, H=H, W=W, d_ffn=dffn, heads=8, seed=seed, shapes=shapes, stride=stride),
                    wsum=np.float64(wsum(w)), out=out[:, ::stride], out_checks=checks(out),
                    h_attn_checks=checks(ha), w_attn_checks=checks(wa), pos=pos[0, :, ::stride, ::stride],
                    pos_checks=checks(pos))
        if ha.numel() < 300000:
            arrs.update(h_attn=ha, w_attn=wa)
        else:
            arrs.update(h_attn=ha[::64, ::16], w_attn=wa[::64, ::16])
        save(f"g2_axial_B{B}_T{T}_C{C}_H{H}_W{W}", **arrs)

    # G2b: the full T*H*W trajectory layer (temporal_attn_type="trajectory", WC/temporal_attention.py:103-155)
    for (B, T, C, H, W, dffn) in [(1, 2, 64, 6, 5, 128), (1, 3, 256, 12, 16, 512), (2, 2, 256, 20, 24, 256)]:
        layer = ta.TemporalTrajectoryAttentionLayer(d_model=C, d_ffn=dffn, n_heads=8).eval()
        seed = 2500 + T * 100 + H
        shapes, w = load_random(layer, seed)
        g = torch.Generator().manual_seed(seed + 1)
        x = torch.randn(B, T, C, H, W, generator=g)
        src = x.permute(0, 1, 3, 4, 2).reshape(B * T, H * W, C).contiguous()
        pos = pe.PositionEmbeddingSine3D(C // 2, normalize=True)(x, fmt="btchw").permute(0, 1, 3, 4, 2).contiguous()
        out, _, _ = layer(src, pos)
        stride = 1 if out.numel() < 200000 else 3
        save(f"g2b_traj_layer_B{B}_T{T}_C{C}_H{H}_W{W}", meta=meta(B=B, T=T, C=C, H=H, W=W, d_ffn=dffn, heads=8, seed=seed, shapes=shapes, stride=stride),
             wsum=np.float64(wsum(w)), out=out[:, ::stride], out_checks=checks(out))
    if "--traj-layer-only" in sys.argv:
        return

    B, T, C, H, W = 2, 2, 64, 6, 5
    enc = ta.TemporalEncoder(d_model=C, d_ffn=128, n_heads=8, temporal_attn_type="axial-trajectory", num_temporal_layer=2).eval()
    shapes, w = load_random(enc, 4000)
    g = torch.Generator().manual_seed(4001)
    x = torch.randn(B, T, C, H, W, generator=g)
    src = x.permute(0, 1, 3, 4, 2).reshape(B * T, H * W, C).contiguous()
    pos = pe.PositionEmbeddingSine3D(C // 2, normalize=True)(x, fmt="btchw").permute(0, 1, 3, 4, 2).contiguous()
    out, ha, wa = enc(src, pos)
    save("g4_encoder_B2_T2_C64_H6_W5", meta=meta(B=B, T=T, C=C, H=H, W=W, d_ffn=128, heads=8, seed=4000, layers=2, shapes=shapes),
         wsum=np.float64(wsum(w)), out=out, h_attn=ha, w_attn=wa)

    # G5: cross-clip pieces and the full module
    for (Bv, Q, Tc) in [(2, 16, 3), (1, 16, 4)]:
        lay = cc.TrajectoryAttentionLayer(d_model=256, nhead=8).eval()
        shapes, w = load_random(lay, 5000 + Tc)
        g = torch.Generator().manual_seed(5100 + Tc)
        x = torch.randn(Bv, Tc * Q, 256, generator=g)
        out = lay(x, seq_len=Q, num_frames=Tc)
        save(f"g5_cc_trajlayer_B{Bv}_Q{Q}_Tc{Tc}", meta=meta(B=Bv, Q=Q, Tc=Tc, seed=5000 + Tc, shapes=shapes),
             wsum=np.float64(wsum(w)), out=out)

        a = cc.ASPP(256, 256, [3, 3, 3], [1, 2, 3], 0.0, "ln").eval()
        shapes, w = load_random(a, 5200 + Tc)
        x = torch.randn(Bv * Q, 256, Tc, generator=g)
        out = a(x)
        save(f"g5_cc_aspp_ln_BQ{Bv * Q}_Tc{Tc}", meta=meta(BQ=Bv * Q, Tc=Tc, seed=5200 + Tc, shapes=shapes),
             wsum=np.float64(wsum(w)), x=x, out=out)

    for (Bv, Q, Tc, V, Hh, Ww, layers, K, full) in [(1, 16, 3, 2, 8, 8, 2, 19, True), (1, 16, 4, 2, 8, 8, 2, 19, True),
                                                    (1, 128, 4, 4, 64, 64, 4, 124, False)]:
        m = cc.CrossClipTrackingModule(num_layers=layers, num_classes=K, attn_drop=0.0, aspp_drop=0.0,
                                       kernel_sizes=[3, 3, 3], atrous_rates=[1, 2, 3], norm_fn="ln",
                                       num_clip_frames=V).eval()
        seed = 6000 + Tc * 10 + layers
        shapes, w = load_random(m, seed)
        g = torch.Generator().manual_seed(seed + 1)
        cq = torch.randn(Bv, Q, Tc, 256, generator=g)
        pf = torch.nn.functional.normalize(torch.randn(Bv, 128, Tc * V, Hh, Ww, generator=g), dim=1)
        out = m(cq, pf)
        arrs = dict(meta=meta(B=Bv, Q=Q, Tc=Tc, V=V, H=Hh, W=Ww, layers=layers, num_classes=K, seed=seed, shapes=shapes),
                    wsum=np.float64(wsum(w)), pred_logits=out["pred_logits"], masks_checks=checks(out["pred_masks"]))
        if full:
            arrs.update(pred_masks=out["pred_masks"], aux0_logits=out["aux_outputs"][0]["pred_logits"],
                        aux0_masks=out["aux_outputs"][0]["pred_masks"])
        else:
            arrs.update(pred_masks=out["pred_masks"][:, ::8, :, ::8, ::8])
        save(f"g5_cc_module_Q{Q}_Tc{Tc}_V{V}_H{Hh}_L{layers}", **arrs)


if __name__ == "__main__":
    main()
