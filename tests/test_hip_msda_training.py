"""GPU: backward of the multi-scale deformable attention op through the C-ABI (axvs_msda_core_bwd = the reference extension's
ms_deform_attn_backward) against the reference-autograd fixtures (tests/golden/g14_*, oracle/gen_golden_msda_bwd.py), the autograd
Function that binds forward + backward, and the module / encoder layer in train() mode against autograd on the float64 oracle."""
import pytest
import torch

import __graft_entry__ as ge
import axvs_oracle as orc
from golden_util import MSDA_BWD, load, msda_core_inputs, rel_err, rel_l2, t

pytestmark = pytest.mark.gpu

TOL = 1e-5   # fp32 arithmetic throughout (grad_value: atomic adds, summation order not fixed)


@pytest.fixture(scope="module", autouse=True)
def built():
    ge.build()
    assert torch.cuda.is_available()


@pytest.mark.parametrize("name", MSDA_BWD)
def test_msda_core_backward_against_reference_autograd(name):
    import axial_vs_amd as ax
    z, m = load(name)
    value, loc, aw = msda_core_inputs(m)
    gv, gl, ga = ax.ms_deform_attn_backward(value.cuda(), m["shapes"], None, loc.cuda(), aw.cuda(), t(z["grad_output"]).cuda())
    e = dict(value=rel_err(gv.cpu(), t(z["grad_value"])), loc=rel_err(gl.cpu(), t(z["grad_sampling_loc"])),
             attn=rel_err(ga.cpu(), t(z["grad_attn_weight"])), value_l2=rel_l2(gv.cpu(), t(z["grad_value"])),
             loc_l2=rel_l2(gl.cpu(), t(z["grad_sampling_loc"])))
    print(f"{name}: {e}")
    assert max(e.values()) < TOL, e
    # the autograd Function: same numbers through torch.autograd, forward against the fixture's output
    v, l, a = (x.cuda().requires_grad_(True) for x in (value, loc, aw))
    out = ax.MSDeformAttnFunction.apply(v, torch.as_tensor(m["shapes"]), None, l, a, 64)
    assert rel_err(out.detach().cpu(), t(z["out"])) < TOL
    out.backward(t(z["grad_output"]).cuda())
    assert rel_err(v.grad.cpu(), t(z["grad_value"])) < TOL and rel_err(l.grad.cpu(), t(z["grad_sampling_loc"])) < TOL
    assert rel_err(a.grad.cpu(), t(z["grad_attn_weight"])) < TOL


def _layer_weights(C, F, M, L, P, seed):
    shapes = {"self_attn.sampling_offsets.weight": (M * L * P * 2, C), "self_attn.sampling_offsets.bias": (M * L * P * 2,),
              "self_attn.attention_weights.weight": (M * L * P, C), "self_attn.attention_weights.bias": (M * L * P,),
              "self_attn.value_proj.weight": (C, C), "self_attn.value_proj.bias": (C,), "self_attn.output_proj.weight": (C, C),
              "self_attn.output_proj.bias": (C,), "norm1.weight": (C,), "norm1.bias": (C,), "linear1.weight": (F, C), "linear1.bias": (F,),
              "linear2.weight": (C, F), "linear2.bias": (C,), "norm2.weight": (C,), "norm2.bias": (C,)}
    w = orc.random_weights(shapes, seed)
    w["self_attn.sampling_offsets.bias"] = w["self_attn.sampling_offsets.bias"] * 20.0      # offsets of a few pixels
    return w


@pytest.mark.parametrize("N,C,shapes,mask", [(2, 256, [(16, 12), (8, 6), (4, 3)], True), (1, 64, [(6, 5), (3, 3)], False)])
def test_msda_encoder_layer_trains(N, C, shapes, mask):
    """MSDeformAttnTransformerEncoderLayer in train() mode (dropout 0): output, input gradient and every parameter gradient against
    autograd on the float64 oracle layer; eval() of the same layer stays on the fused inference kernels and agrees to their bar."""
    import axial_vs_amd as ax
    M, P, F = 8, 4, 2 * C
    L, S = len(shapes), sum(h * w for h, w in shapes)
    w = _layer_weights(C, F, M, L, P, 71)
    g = torch.Generator().manual_seed(72)
    src, pos = torch.randn(N, S, C, generator=g), torch.randn(N, S, C, generator=g) * 0.5
    ref_pts = torch.rand(N, S, L, 2, generator=g)
    pad = (torch.rand(N, S, generator=g) < 0.1) if mask else None
    d_out = torch.randn(N, S, C, generator=g)
    wd = {k: v.double().requires_grad_(True) for k, v in w.items()}
    sd = src.double().requires_grad_(True)
    ref = orc.msda_encoder_layer(sd, pos.double(), ref_pts.double(), shapes, wd, M, L, P, padding_mask=pad)
    ref.backward(d_out.double())
    layer = ax.MSDeformAttnTransformerEncoderLayer(C, F, dropout=0.0, n_levels=L, n_heads=M, n_points=P)
    layer.load_state_dict(w, strict=True)
    layer = layer.cuda().train()
    s = src.cuda().requires_grad_(True)
    out = layer(s, pos.cuda(), ref_pts.cuda(), torch.as_tensor(shapes).cuda(), None, pad.cuda() if mask else None)
    out.backward(d_out.cuda())
    e = dict(out=rel_err(out.detach().cpu(), ref.detach()), d_src=rel_err(s.grad.cpu(), sd.grad))
    scale = max(float(v.grad.norm()) for v in wd.values())
    pe = {k: float((p.grad.cpu().double() - wd[k].grad).norm() / max(float(wd[k].grad.norm()), 1e-3 * scale)) for k, p in layer.named_parameters()}
    print(f"N={N} C={C}: {e} worst parameter gradient {max(pe, key=pe.get)} {max(pe.values()):.2e}")
    assert max(e.values()) < 1e-4 and max(pe.values()) < 1e-4, (e, pe)
    with torch.no_grad():
        out_eval = layer.eval()(src.cuda(), pos.cuda(), ref_pts.cuda(), torch.as_tensor(shapes).cuda(), None, pad.cuda() if mask else None)
    assert rel_err(out_eval.cpu(), ref.detach()) < 1e-3


@pytest.mark.parametrize("name", ["g8_pixel_decoder_T2_S2", "g8_pixel_decoder_T3_S1", "g8_pixel_decoder_T2_S2_temporal_only"])
def test_within_clip_module_trains(name):
    """WithinClipTrackingModule in train() mode (dropout 0): forward_features equals the reference decoder's fixture outputs at fp32
    accuracy (the training tiers are fp32), and the gradients of the backbone maps and of every parameter equal autograd on the
    float64 oracle decoder -- the whole within-clip stage of the reference trains on this package's kernels (deformable attention:
    HIP forward / backward of the native op; axial-trajectory layers: their training tier; 1x1 conv + GroupNorm: the library's training tier since round 6,
    axial_vs_amd.glue_training)."""
    from golden_util import weights
    from test_cabi_cpu import _decoder_from_meta
    z, m = load(name)
    w = weights(z, m)
    mod = _decoder_from_meta(m)
    mod.within_clip_tracking_module.load_state_dict(w, strict=True)
    mod = mod.cuda().train()
    g = torch.Generator().manual_seed(m["seed"] + 1)
    feats = {k: torch.randn(m["B"] * m["T"], m["chans"][k], *m["sizes"][k], generator=g) for k in m["chans"]}
    d_out = {k: torch.randn(feats[k].shape, generator=g) for k in feats}
    fin = {k: v.cuda().requires_grad_(True) for k, v in feats.items()}
    out, _, _ = mod.forward_features(dict(fin))
    for k in m["chans"]:
        e = rel_err(out[k].detach().cpu(), t(z["out_" + k]))
        print(f"{name} {k}: train-mode forward vs the reference decoder {e:.2e}")
        assert e < 1e-4, k
    sum((out[k] * d_out[k].cuda()).sum() for k in out).backward()
    # the same under float64 autograd on the oracle decoder
    wd = {k: v.double().requires_grad_(True) for k, v in w.items()}
    fd = {k: v.double().requires_grad_(True) for k, v in feats.items()}
    ref = orc.pixel_decoder(fd, wd, ["res3", "res4", "res5"], ["res4", "res5"], m["stages"], m["temporal_per_stage"], num_clip_frames=m["T"], B=m["B"],
                            with_spatial=not m.get("temporal_only"))
    sum((ref[k] * d_out[k].double()).sum() for k in ref).backward()
    e = {k: rel_err(fin[k].grad.cpu(), fd[k].grad) for k in fin}
    scale = max(float(v.grad.norm()) for v in wd.values() if v.grad is not None)
    pe = {k: float((p.grad.cpu().double() - wd[k].grad).norm() / max(float(wd[k].grad.norm()), 1e-3 * scale))
          for k, p in mod.within_clip_tracking_module.named_parameters() if wd[k].grad is not None}
    worst = max(pe, key=pe.get)
    print(f"{name}: feature gradients {e}, worst parameter gradient {worst} {pe[worst]:.2e} ({len(pe)} parameters)")
    # where a level passes through no layer (temporal-only decoder: res3) the GroupNorm bias of its input projection feeds a second GroupNorm that
    # removes most of it -- a gradient that nearly cancels, left with fp32 rounding (torch's own kernels: 1.5e-4 of the floor; the library's input-gradient
    # GEMM of the projections runs three-piece operands for this reason)
    glue = {k: v for k, v in pe.items() if k.startswith(("input_proj", "output_proj"))}
    ours = {k: v for k, v in pe.items() if k not in glue}
    assert max(e.values()) < 1e-4 and max(ours.values()) < 1e-4 and max(glue.values()) < 5e-4, (e, worst, pe[worst])
    assert all(p.grad is not None for p in mod.parameters())


from golden_util import TL_PLUGIN, tl_plugin_case  # noqa: E402


@pytest.mark.parametrize("name", TL_PLUGIN)
def test_tube_link_plugin_trains(name):
    """MultiScaleDeformableAxialTrajectoryAttention (TL ...pixel_decoder.py:393-638) in train() mode, dropout 0: the output equals the
    reference class's fixture output at fp32 accuracy and the gradients of query and of every parameter (gamma, the temporal encoder,
    the four projections) equal autograd on the float64 oracle."""
    import axial_vs_amd as ax
    z, m = load(name)
    w, q, qp, pos3d, ref, km = tl_plugin_case(z, m)
    shapes = [tuple(s) for s in m["shapes"]]
    mod = ax.MultiScaleDeformableAxialTrajectoryAttention(
        embed_dims=256, num_heads=8, num_levels=len(shapes), num_temporal_levels=m["temporal_levels"], num_temporal_layers=m["layers"],
        num_temporal_dim=m["d_ffn"], num_points=4, dropout=0.0, batch_first=m["batch_first"], skip_connect=m["skip_connect"])
    mod.load_state_dict(w, strict=True)
    mod = mod.cuda().train()
    perm = (lambda x: x) if m["batch_first"] else (lambda x: x.permute(1, 0, 2))
    ss = torch.as_tensor(shapes, dtype=torch.long, device="cuda")
    qc = q.cuda().requires_grad_(True)
    out = mod(query=perm(qc), query_pos=perm(qp.cuda()), query_pos3d=[p.cuda() for p in pos3d],
              key_padding_mask=km.cuda() if km is not None else None, reference_points=ref.cuda(), spatial_shapes=ss)
    out_bf = out if m["batch_first"] else perm(out)
    e = rel_err(out_bf.detach().cpu()[:, ::m["stride"]], t(z["out"]))
    g = torch.Generator().manual_seed(5)
    d_out = torch.randn(out_bf.shape, generator=g)
    (out_bf * d_out.cuda()).sum().backward()
    wd = {k: v.double().requires_grad_(True) for k, v in w.items()}
    qd = q.double().requires_grad_(True)
    refo = orc.tl_plugin_attention(qd, qp.double(), [p.double() for p in pos3d], ref.double(), shapes, wd, 8, 4, m["temporal_levels"], m["layers"],
                                   skip_connect=m["skip_connect"], key_padding_mask=km)
    (refo * d_out.double()).sum().backward()
    # Bilinear sampling has a kink where a sampling coordinate is an integer (the derivative wrt the location jumps from one pixel
    # pair to the next): a coordinate within fp32 rounding of an integer floors differently in fp32 and float64, and the query row
    # that owns the sample then differs by O(1e-3) of the gradient scale -- the same kind of tie as a ReLU pre-activation at zero
    # (test_relu_ties_bound_fp32_gradient_parity).  The reference points are pixel centres, so loc * W - 0.5 = integer + offset:
    # rows with a coordinate within 1e-5 of an integer are left out of the comparison (the rows that differed: 1 of 3528, 1 of 5120).
    with torch.no_grad():
        qq = (q + qp).double()
        off = (qq @ w["sampling_offsets.weight"].double().T + w["sampling_offsets.bias"].double()).reshape(q.shape[0], q.shape[1], 8, len(shapes), 4, 2)
        wh = torch.tensor([[ww_, h_] for h_, ww_ in shapes], dtype=torch.float64)
        coord = (ref.double()[:, :, None, :, None, :] + off / wh[None, None, None, :, None, :]) * wh[None, None, None, :, None, :] - 0.5
        frac = coord - torch.floor(coord)
        kink = ((frac < 1e-5) | (frac > 1 - 1e-5)).flatten(2).any(-1)          # [bs, num_query]
    print(f"{name}: {int(kink.sum())} of {kink.numel()} query rows sample within 1e-5 of a pixel boundary")
    keep = ~kink
    eq = float((qc.grad.cpu().double() - qd.grad)[keep].abs().max() / qd.grad.abs().max())
    assert rel_l2(qc.grad.cpu()[keep], qd.grad[keep]) < 1e-4 and int(kink.sum()) <= kink.numel() // 100
    scale = max(float(v.grad.norm()) for v in wd.values() if v.grad is not None)
    pe = {k: float((p.grad.cpu().double() - wd[k].grad).norm() / max(float(wd[k].grad.norm()), 1e-3 * scale))
          for k, p in mod.named_parameters() if wd[k].grad is not None}
    worst = max(pe, key=pe.get)
    print(f"{name}: forward vs the reference class {e:.2e}, d_query {eq:.2e}, worst parameter gradient {worst} {pe[worst]:.2e}")
    # (the kink rows still count in the sums over rows that make the sampling_offsets gradients: one such row moves them by a few 1e-4)
    off_bound = 1e-3 if int(kink.sum()) else 1e-4
    assert e < 1e-4 and eq < 1e-4
    assert all(v < (off_bound if k.startswith("sampling_offsets") else 1e-4) for k, v in pe.items()), (worst, pe[worst])
