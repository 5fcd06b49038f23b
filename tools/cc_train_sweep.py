"""Random-shape sweep of the cross-clip module's training tier against autograd on the float64 oracle restatement (itself pinned to the
reference by tests/golden/g13_*): batch > 1, 1..13 clips, odd pixel counts, 1..3 layers, both dropouts.
    python tools/cc_train_sweep.py [n] [seed]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import axvs_oracle as orc
import axial_vs_amd as ax
from golden_util import rel_err

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=rng))
worst, fails = 0.0, 0
for it in range(n):
    B, Q, Tc, V, H, W, nl, K = ri(1, 3), 8 * ri(1, 5), ri(1, 13), ri(1, 3), ri(2, 9), ri(2, 11), ri(1, 3), ri(2, 20)
    p_attn, p_aspp = [0.0, 0.1, 0.3][ri(0, 2)], [0.0, 0.2][ri(0, 1)]
    seed = 1000 + it
    w = orc.random_weights(orc.cc_module_param_shapes(nl, K), seed)
    g = torch.Generator().manual_seed(seed + 1)
    cq = torch.randn(B, Q, Tc, 256, generator=g)
    pf = torch.nn.functional.normalize(torch.randn(B, 128, Tc * V, H, W, generator=g), dim=1)
    d_l = [torch.randn(1, Q, K + 1, generator=g) for _ in range(nl)]
    d_m = [torch.randn(B, Q, Tc * V, H, W, generator=g) * 0.05 for _ in range(nl)]
    wd = {k: v.double().requires_grad_("running" not in k) for k, v in w.items()}
    qd = cq.double().requires_grad_(True)
    rl, rm, _ = orc.cc_module_train(qd, pf.double(), wd, nl, V, (1, 2, 3), p_attn, p_aspp, seed)
    (sum((a * b.double()).sum() for a, b in zip(rl, d_l)) + sum((a * b.double()).sum() for a, b in zip(rm, d_m))).backward()
    mod = ax.CrossClipTrackingModule(num_layers=nl, num_classes=K, attn_drop=p_attn, aspp_drop=p_aspp, kernel_sizes=[3, 3, 3], atrous_rates=[1, 2, 3],
                                     norm_fn="ln", num_clip_frames=V)
    sd = mod.state_dict(); sd.update(w); mod.load_state_dict(sd, strict=True)
    mod = mod.cuda().train(); mod.dropout_seed = seed
    q = cq.cuda().requires_grad_(True)
    out = mod(q, pf.cuda())
    logits = [a["pred_logits"] for a in out["aux_outputs"]] + [out["pred_logits"]]
    masks = [a["pred_masks"] for a in out["aux_outputs"]] + [out["pred_masks"]]
    (sum((a * b.cuda()).sum() for a, b in zip(logits, d_l)) + sum((a * b.cuda()).sum() for a, b in zip(masks, d_m))).backward()
    e = max(rel_err(torch.stack([x.detach() for x in logits]).cpu(), torch.stack([x.detach() for x in rl])),
            rel_err(torch.stack([x.detach() for x in masks]).cpu(), torch.stack([x.detach() for x in rm])), rel_err(q.grad.cpu(), qd.grad))
    names = [k for k, v in wd.items() if v.requires_grad]
    scale = max(float(wd[k].grad.norm()) for k in names)
    grads = {k: v.grad for k, v in mod.named_parameters()}
    pe = max(float((grads[k].cpu().double() - wd[k].grad).norm() / max(float(wd[k].grad.norm()), 1e-3 * scale)) for k in names)
    ok = e < 1e-4 and pe < 1e-4
    worst = max(worst, e, pe)
    fails += not ok
    print(f"{'ok  ' if ok else 'FAIL'} B={B} Q={Q} Tc={Tc} V={V} H={H} W={W} (P={V*H*W}) layers={nl} K={K} p=({p_attn},{p_aspp}): outputs/d_query {e:.1e} params {pe:.1e}", flush=True)
print(f"{n} shapes, {fails} failures, worst {worst:.2e}")
