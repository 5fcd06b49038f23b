"""GPU: the pixel decoder's 1x1 convolution + GroupNorm projections in train() mode (WC/msdeformattn.py:349-375 under autograd) through the library's training
tier (axvs_conv1x1_gn_train_fwd / _bwd, axial_vs_amd.glue_training) against torch autograd of the same nn.Conv2d + nn.GroupNorm in float64 on the CPU."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize("N,Cin,Cout,H,W,direction", [(2, 64, 256, 5, 7, "in"), (3, 256, 96, 9, 8, "out"), (4, 768, 256, 16, 16, "in"), (1, 256, 192, 64, 64, "out"),
                                                       (2, 2048, 256, 25, 43, "in")])
def test_conv1x1_groupnorm_training_tier_against_torch_autograd_float64(N, Cin, Cout, H, W, direction):
    from axial_vs_amd.glue_training import conv_gn_train
    g = torch.Generator().manual_seed(1000 + Cin + H)
    conv = torch.nn.Conv2d(Cin, Cout, 1)
    gn = torch.nn.GroupNorm(32, Cout)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * (1.5 / Cin ** 0.5))
        conv.bias.copy_(torch.randn(Cout, generator=g) * 0.3)
        gn.weight.copy_(torch.rand(Cout, generator=g) + 0.5)
        gn.bias.copy_(torch.randn(Cout, generator=g) * 0.2)
    x = torch.randn(N, Cin, H, W, generator=g) * 1.3 + 0.2
    d_out = torch.randn(N, Cout, H, W, generator=g)
    # float64 reference (NCHW throughout)
    c64, g64 = torch.nn.Conv2d(Cin, Cout, 1).double(), torch.nn.GroupNorm(32, Cout).double()
    c64.load_state_dict({k: v.double() for k, v in conv.state_dict().items()})
    g64.load_state_dict({k: v.double() for k, v in gn.state_dict().items()})
    x64 = x.double().requires_grad_(True)
    ref = g64(c64(x64))
    ref.backward(d_out.double())
    # library
    conv, gn = conv.cuda(), gn.cuda()
    if direction == "in":          # input_proj: NCHW map -> token rows
        xg = x.cuda().requires_grad_(True)
        out = conv_gn_train(xg, conv, gn, out_layout="tokens")
        assert out.shape == (N, H * W, Cout)
        out.backward(d_out.flatten(2).transpose(1, 2).contiguous().cuda())
        got_out = out.detach().transpose(1, 2).reshape(N, Cout, H, W)
        got_dx = xg.grad
    else:                          # output_proj: token rows -> NCHW map
        xg = x.flatten(2).transpose(1, 2).contiguous().cuda().requires_grad_(True)
        out = conv_gn_train(xg, conv, gn, out_layout="nchw", hw=(H, W))
        assert out.shape == (N, Cout, H, W)
        out.backward(d_out.cuda())
        got_out = out.detach()
        got_dx = xg.grad.transpose(1, 2).reshape(N, Cin, H, W)
    torch.cuda.synchronize()
    errs = {"out": _rel(got_out, ref), "d_x": _rel(got_dx, x64.grad), "d_conv_w": _rel(conv.weight.grad, c64.weight.grad), "d_conv_b": _rel(conv.bias.grad, c64.bias.grad),
            "d_gn_w": _rel(gn.weight.grad, g64.weight.grad), "d_gn_b": _rel(gn.bias.grad, g64.bias.grad)}
    print(f"[conv1x1+GN train] N={N} Cin={Cin} Cout={Cout} {H}x{W} {direction}: " + "  ".join(f"{k} {v:.1e}" for k, v in errs.items()))
    assert errs["out"] < 5e-6, errs
    assert max(errs.values()) < 1e-4, errs
    # deterministic: the same call gives the same bits
    xg2 = xg.detach().clone().requires_grad_(True)
    conv.zero_grad(); gn.zero_grad()
    out2 = conv_gn_train(xg2, conv, gn, out_layout="tokens" if direction == "in" else "nchw", hw=(H, W))
    assert torch.equal(out2.detach(), out.detach())


def test_within_clip_module_train_mode_runs_no_torch_convolution():
    """The train() forward of the pixel decoder calls the library for its projections: no aten convolution / native_group_norm kernel is recorded."""
    import axial_vs_amd as ax
    from test_cabi_cpu import _decoder_from_meta
    from golden_util import load, weights
    z, m = load("g8_pixel_decoder_T2_S2")
    mod = _decoder_from_meta(dict(m), cross_clip_training=True)
    mod.within_clip_tracking_module.load_state_dict(weights(z, m), strict=True)
    mod = mod.cuda().train()
    g = torch.Generator().manual_seed(m["seed"] + 1)
    feats = {k: torch.randn(m["B"] * m["T"], m["chans"][k], *m["sizes"][k], generator=g).cuda().requires_grad_(True) for k in m["chans"]}
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU]) as prof:
        out, _, _ = mod.forward_features(dict(feats))
        sum(o.float().sum() for o in out.values()).backward()
    torch.cuda.synchronize()
    names = {e.key for e in prof.key_averages()}
    assert not any("convolution" in n or "group_norm" in n for n in names), sorted(n for n in names if "conv" in n or "norm" in n)
    assert all(f.grad is not None and torch.isfinite(f.grad).all() for f in feats.values())
