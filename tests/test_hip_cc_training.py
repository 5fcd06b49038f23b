"""GPU: the training tier of the cross-clip tracking module -- forward + backward through the C-ABI -- against the
reference-autograd fixtures (tests/golden/g13_*, oracle/gen_golden_cc_train.py) and, at BASELINE config 4's size, against autograd on
the float64 oracle restatement."""
import numpy as np
import pytest
import torch

import __graft_entry__ as ge
import axvs_oracle as orc
from golden_util import CC_TRAIN, load, rel_err, rel_l2, t, train_grad_errors, weights

pytestmark = pytest.mark.gpu

TOL = 1e-4   # fp32 activations, split-bf16 GEMMs with fp32 accuracy in the forward: observed ~1e-6 .. 2e-5


@pytest.fixture(scope="module", autouse=True)
def built():
    ge.build()
    assert torch.cuda.is_available()


def make_module(m, w, seed):
    import axial_vs_amd as ax
    mod = ax.CrossClipTrackingModule(num_layers=m["layers"], num_classes=m["num_classes"], attn_drop=m["p_attn_drop"], aspp_drop=m["p_aspp_drop"],
                                     kernel_sizes=[3, 3, 3], atrous_rates=[1, 2, 3], norm_fn="ln", num_clip_frames=m["V"])
    sd = mod.state_dict()
    sd.update(w)
    mod.load_state_dict(sd, strict=True)
    mod = mod.cuda().train()
    mod.dropout_seed = seed
    return mod


def inputs(m):
    g = torch.Generator().manual_seed(m["seed"] + 1)
    cq = torch.randn(m["B"], m["Q"], m["Tc"], 256, generator=g)
    pf = torch.nn.functional.normalize(torch.randn(m["B"], 128, m["Tc"] * m["V"], m["H"], m["W"], generator=g), dim=1)
    return cq, pf


def run(mod, cq, pf, d_logits, d_masks):
    q = cq.float().cuda().requires_grad_(True)
    out = mod(q, pf.float().cuda())
    logits = [a["pred_logits"] for a in out["aux_outputs"]] + [out["pred_logits"]]
    masks = [a["pred_masks"] for a in out["aux_outputs"]] + [out["pred_masks"]]
    assert all(x.is_cuda and x.requires_grad for x in logits + masks)       # the training branch keeps its outputs on the GPU (CC:53-57)
    loss = sum((a * b.cuda()).sum() for a, b in zip(logits, d_logits)) + sum((a * b.cuda()).sum() for a, b in zip(masks, d_masks))
    loss.backward()
    return (torch.stack([x.detach() for x in logits]).cpu(), torch.stack([x.detach() for x in masks]).cpu(), q.grad.cpu(),
            {k: v.grad.cpu() for k, v in mod.named_parameters()})


@pytest.mark.parametrize("name", CC_TRAIN)
def test_cc_training_tier_against_reference_autograd(name):
    z, m = load(name)
    w = weights(z, m)
    cq, pf = inputs(m)
    mod = make_module(m, w, m["dropout_seed"])
    logits, masks, d_cq, grads = run(mod, cq, pf, list(t(z["d_logits"])), list(t(z["d_masks"])))
    e = dict(logits=rel_err(logits, t(z["logits"])), masks=rel_err(masks, t(z["masks"])), d_clip_query=rel_err(d_cq, t(z["d_clip_query"])),
             logits_l2=rel_l2(logits, t(z["logits"])), masks_l2=rel_l2(masks, t(z["masks"])), d_clip_query_l2=rel_l2(d_cq, t(z["d_clip_query"])))
    ge_ = train_grad_errors(z, grads)
    worst = max(ge_, key=ge_.get)
    print(f"{name}: {e} worst parameter gradient {worst} {ge_[worst]:.2e}")
    assert max(e.values()) < TOL, e
    assert max(ge_.values()) < TOL, ge_
    for k, p in mod.named_parameters():
        assert p.grad.shape == p.shape and p.grad.dtype == p.dtype
    # the forward updated the running statistics of the four BatchNorm sites like the reference's (once per layer, momentum 0.01)
    for k, b in mod.named_buffers():
        ref = t(z["buf." + k])
        if b.dtype.is_floating_point:
            assert rel_err(b.cpu(), ref) < TOL, k
        else:
            assert int(b) == int(ref), k


def test_cc_training_dropout_is_a_function_of_the_seed_and_eval_tier_agrees():
    z, m = load(CC_TRAIN[1])           # the fixture with both dropouts on
    w = weights(z, m)
    cq, pf = inputs(m)
    dl, dm = list(t(z["d_logits"])), list(t(z["d_masks"]))
    a = run(make_module(m, w, 7), cq, pf, dl, dm)
    b = run(make_module(m, w, 7), cq, pf, dl, dm)
    c = run(make_module(m, w, 8), cq, pf, dl, dm)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and all(torch.equal(a[3][k], b[3][k]) for k in a[3])
    assert not torch.equal(a[1], c[1])


def test_cc_training_at_baseline_config_4_vs_float64_oracle():
    """BASELINE config 4 (4 clips x 4 frames, 64 x 64 pixel features, 128 queries, 4 layers): outputs and gradients against
    torch.autograd on the float64 oracle restatement of the module's training branch."""
    B, Q, Tc, V, H, W, nl, K = 1, 128, 4, 4, 64, 64, 4, 124
    shapes = orc.cc_module_param_shapes(nl, K)
    w = orc.random_weights(shapes, 77)
    g = torch.Generator().manual_seed(78)
    cq = torch.randn(B, Q, Tc, 256, generator=g)
    pf = torch.nn.functional.normalize(torch.randn(B, 128, Tc * V, H, W, generator=g), dim=1)
    d_logits = [torch.randn(1, Q, K + 1, generator=g) for _ in range(nl)]
    d_masks = [torch.randn(B, Q, Tc * V, H, W, generator=g) * 0.01 for _ in range(nl)]
    m = dict(layers=nl, num_classes=K, p_attn_drop=0.1, p_aspp_drop=0.1, V=V)
    seed = 31337
    wd = {k: v.double().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in w.items()}
    qd = cq.double().requires_grad_(True)
    ref_logits, ref_masks, ref_stats = orc.cc_module_train(qd, pf.double(), wd, nl, V, [1, 2, 3], m["p_attn_drop"], m["p_aspp_drop"], seed)
    loss = sum((a * b.double()).sum() for a, b in zip(ref_logits, d_logits)) + sum((a * b.double()).sum() for a, b in zip(ref_masks, d_masks))
    loss.backward()
    mod = make_module(m, w, seed)
    logits, masks, d_cq, grads = run(mod, cq, pf, d_logits, d_masks)
    rl, rm = torch.stack([x.detach() for x in ref_logits]), torch.stack([x.detach() for x in ref_masks])
    e = dict(logits=rel_err(logits, rl), masks=rel_err(masks, rm), d_clip_query=rel_err(d_cq, qd.grad), logits_l2=rel_l2(logits, rl),
             masks_l2=rel_l2(masks, rm), d_clip_query_l2=rel_l2(d_cq, qd.grad))
    names = [k for k, v in wd.items() if v.requires_grad]
    scale = max(float(wd[k].grad.norm()) for k in names)
    pe = {k: float((grads[k].double() - wd[k].grad).norm() / max(float(wd[k].grad.norm()), 1e-3 * scale)) for k in names}
    worst = max(pe, key=pe.get)
    print(f"cfg 4: {e} worst parameter gradient {worst} {pe[worst]:.2e}")
    assert max(e.values()) < TOL, e
    assert max(pe.values()) < TOL, pe


def test_cc_training_rejects_what_it_does_not_build():
    import axial_vs_amd as ax
    mod = ax.CrossClipTrackingModule(num_layers=1, num_classes=3, attn_drop=0.0, aspp_drop=0.0, kernel_sizes=[3, 3, 3], atrous_rates=[1, 2, 3],
                                     norm_fn="ln", num_clip_frames=1).cuda().train()
    with pytest.raises(RuntimeError, match="multiple of 8"):                      # the number of queries
        mod(torch.randn(1, 12, 2, 256, device="cuda"), torch.randn(1, 128, 2, 4, 8, device="cuda"))
    with pytest.raises(RuntimeError, match="16 clips"):
        mod(torch.randn(1, 8, 17, 256, device="cuda"), torch.randn(1, 128, 17, 4, 8, device="cuda"))
    with pytest.raises(NotImplementedError, match="panoptic_features"):
        mod(torch.randn(1, 16, 2, 256, device="cuda"), torch.randn(1, 128, 2, 4, 8, device="cuda", requires_grad=True))


def test_cc_training_amp_autocast_and_grad_scaling():
    """The shipped config trains with AMP (SOLVER.AMP.ENABLED, maxtron_cc_r50.yaml:98-99): under torch.autocast with half-precision
    inputs the module computes in fp32 at its boundary (outputs fp32), hands gradients back in each input's dtype, and is linear in
    the loss scale -- so GradScaler works unchanged."""
    z, m = load(CC_TRAIN[1])
    w = weights(z, m)
    cq, pf = inputs(m)
    dl = [x.cuda() for x in t(z["d_logits"])]
    dm = [x.cuda() for x in t(z["d_masks"])]

    def loss_of(out, scale):
        logits = [a["pred_logits"] for a in out["aux_outputs"]] + [out["pred_logits"]]
        masks = [a["pred_masks"] for a in out["aux_outputs"]] + [out["pred_masks"]]
        return (sum((a * b).sum() for a, b in zip(logits, dl)) + sum((a * b).sum() for a, b in zip(masks, dm))) * scale, logits, masks

    mod = make_module(m, w, 5)
    q32 = cq.cuda().half().float().requires_grad_(True)          # the same fp16-representable inputs, in fp32 without autocast
    p32 = pf.cuda().half().float()
    loss_of(mod(q32, p32), 1.0)[0].backward()
    g32 = {k: v.grad.clone() for k, v in mod.named_parameters()}
    mod2 = make_module(m, w, 5)
    mod2.amp_compute = False                                     # split-precision products under autocast too: the boundary alone is under test
    q16 = cq.cuda().half().requires_grad_(True)
    with torch.autocast(device_type="cuda", dtype=torch.float16):
        loss, logits, masks = loss_of(mod2(q16, pf.cuda().half()), 1024.0)
    assert all(x.dtype == torch.float32 for x in logits + masks)
    loss.backward()
    assert q16.grad.dtype == torch.float16
    for k, v in mod2.named_parameters():
        assert v.grad.dtype == torch.float32
        assert rel_l2(v.grad.cpu() / 1024.0, g32[k].cpu()) < 1e-5 or float(g32[k].norm()) < 1e-4 * max(float(x.norm()) for x in g32.values()), k
    assert rel_l2(q16.grad.float().cpu() / 1024.0, q32.grad.cpu()) < 1e-3      # d_clip_query is returned in fp16


def test_cc_training_amp_runs_16_bit_products():
    """The default under autocast (the shipped config's SOLVER.AMP): the X W^T GEMMs of the module -- q/k/v, projections, ASPP
    branches, heads, the mask einsum -- multiply one fp16 piece per operand, as autocast makes the reference's Linear / Conv / einsum
    do; BatchNorm statistics, softmax, LayerNorm and weight gradients stay fp32 / split precision.  Against the fp32 tier on the
    same inputs: relative L2 at the fp16 level."""
    z, m = load(CC_TRAIN[1])
    w = weights(z, m)
    cq, pf = inputs(m)
    dl = [x.cuda() for x in t(z["d_logits"])]
    dm = [x.cuda() for x in t(z["d_masks"])]

    def run(amp):
        mod = make_module(m, w, 5)
        q = cq.cuda().requires_grad_(True)
        if amp:
            with torch.autocast(device_type="cuda", dtype=torch.float16):
                out = mod(q, pf.cuda())
        else:
            out = mod(q, pf.cuda())
        logits = [a["pred_logits"] for a in out["aux_outputs"]] + [out["pred_logits"]]
        masks = [a["pred_masks"] for a in out["aux_outputs"]] + [out["pred_masks"]]
        ((sum((a * b).sum() for a, b in zip(logits, dl)) + sum((a * b).sum() for a, b in zip(masks, dm))) * 64.0).backward()
        return torch.stack(masks).detach().cpu(), q.grad.cpu() / 64.0, {k: v.grad.cpu() / 64.0 for k, v in mod.named_parameters()}

    mk32, dq32, g32 = run(False)
    mk16, dq16, g16 = run(True)
    scale = max(float(v.norm()) for v in g32.values())
    e_w = max(float((g16[k] - g32[k]).norm()) / max(float(g32[k].norm()), 1e-2 * scale) for k in g32)
    e = dict(masks=rel_l2(mk16, mk32), d_clip_query=rel_l2(dq16, dq32), worst_param_grad=e_w)
    print("cross-clip module under fp16 autocast vs the fp32 tier (relative L2):", {k: f"{v:.2e}" for k, v in e.items()})
    assert 1e-6 < e["masks"] < 1e-2 and e["d_clip_query"] < 5e-2 and e_w < 5e-2, e


@pytest.mark.parametrize("name", ["g6_tl_cc_head_Tc3_Q16_f2_L2", "g6_tl_cc_head_Tc2_Q20_f1_L1", "g6_tl_cc_head_Tc4_Q100_f2_L4"])
def test_tube_link_cross_clip_head_trains(name):
    """Tube-Link's cross-clip head in train() mode, dropouts 0: the layer chain on the library's training tier
    (axvs_cc_layers_train_*), its prediction heads as torch modules.  Outputs of every layer against the reference class's fixtures at
    fp32 accuracy; gradients of the clip queries and of every parameter against autograd on the float64 oracle."""
    import axial_vs_amd as ax
    z, m = load(name)
    w = weights(z, m)
    g = torch.Generator().manual_seed(m["seed"] + 1)
    cq = torch.randn(m["B"], m["Tc"], m["Q"], 256, generator=g)
    mf = torch.nn.functional.normalize(torch.randn(m["B"], m["Tc"] * m["fpc"], m["Cm"], m["h"], m["w"], generator=g), dim=2)
    mod = ax.TubeLinkCrossClipHead(num_classes=m["num_classes"], out_channels=m["Cm"], num_cc_layers=m["layers"], trajectory_drop_out=0.0,
                                   drop_path_prob=0.0)
    mod.load_state_dict(w, strict=True)
    mod = mod.cuda().train()
    q = cq.cuda().requires_grad_(True)
    cls, masks = mod(q, mf.cuda())
    assert rel_err(cls[-1].detach().cpu(), t(z["cls_last"])) < TOL and rel_err(cls[0].detach().cpu(), t(z["cls_first"])) < TOL
    d_cls = [torch.randn(c.shape, generator=g) for c in cls]
    d_masks = [torch.randn(x.shape, generator=g) * 0.05 for x in masks]
    (sum((a * b.cuda()).sum() for a, b in zip(cls, d_cls)) + sum((a * b.cuda()).sum() for a, b in zip(masks, d_masks))).backward()
    wd = {k: v.double().requires_grad_(True) for k, v in w.items()}
    qd = cq.double().requires_grad_(True)
    rc, rm = orc.tl_cross_clip_head(qd, mf.double(), wd, m["layers"])
    (sum((a * b.double()).sum() for a, b in zip(rc, d_cls)) + sum((a * b.double()).sum() for a, b in zip(rm, d_masks))).backward()
    e = dict(masks=rel_err(masks[-1].detach().cpu(), rm[-1].detach()), d_clip_query=rel_err(q.grad.cpu(), qd.grad))
    scale = max(float(v.grad.norm()) for v in wd.values() if v.grad is not None)
    pe = {k: float((p.grad.cpu().double() - wd[k].grad).norm() / max(float(wd[k].grad.norm()), 1e-3 * scale)) for k, p in mod.named_parameters()
          if wd[k].grad is not None}
    worst = max(pe, key=pe.get)
    print(f"{name}: {e} worst parameter gradient {worst} {pe[worst]:.2e} ({len(pe)} parameters)")
    assert max(e.values()) < TOL and max(pe.values()) < TOL
