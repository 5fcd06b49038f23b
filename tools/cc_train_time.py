"""Training step (forward + backward) of CrossClipTrackingModule at BASELINE config 4 through the training tier; --torch: the
same math as torch-eager autograd on this GPU (the oracle restatement in fp32 on the device).
    python tools/cc_train_time.py [steps] [--torch] [--p 0.1] [--shape Q,Tc,V,H,W,layers]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import axvs_oracle as orc  # noqa: E402
import axial_vs_amd as ax  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 10
p = float(sys.argv[sys.argv.index("--p") + 1]) if "--p" in sys.argv else 0.1
B, Q, Tc, V, H, W, nl, K = 1, 128, 4, 4, 64, 64, 4, 124
if "--shape" in sys.argv:          # --shape Q,Tc,V,H,W,layers   e.g. the shipped VIPSeg training setting: 128,12,2,193,337,6
    Q, Tc, V, H, W, nl = (int(v) for v in sys.argv[sys.argv.index("--shape") + 1].split(","))
dev = torch.device("cuda:0")
w = orc.random_weights(orc.cc_module_param_shapes(nl, K), 4)
g = torch.Generator().manual_seed(4)
cq = torch.randn(B, Q, Tc, 256, generator=g).to(dev).requires_grad_(True)
pf = torch.nn.functional.normalize(torch.randn(B, 128, Tc * V, H, W, generator=g), dim=1).to(dev)
d_l = torch.randn(nl, 1, Q, K + 1, generator=g).to(dev)
d_m = (torch.randn(nl, B, Q, Tc * V, H, W, generator=g) * 0.01).to(dev)


def timed(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


if "--torch" in sys.argv:
    wd = {k: v.to(dev).requires_grad_("running" not in k) for k, v in w.items()}
    orig = orc.dropout_keep
    orc.dropout_keep = lambda seed, site, count, pp, dtype=torch.float32: (torch.rand(count, device=dev) >= pp).to(dtype) / (1 - pp) if pp > 0 else torch.ones(count, device=dev, dtype=dtype)

    def step():
        lg, mk, _ = orc.cc_module_train(cq, pf, wd, nl, V, (1, 2, 3), p, p, 1)
        (sum((a * b).sum() for a, b in zip(lg, d_l)) + sum((a * b).sum() for a, b in zip(mk, d_m))).backward()

    def fwd():
        with torch.no_grad():
            orc.cc_module_train(cq, pf, wd, nl, V, (1, 2, 3), p, p, 1)
    print(f"torch eager fp32 (oracle restatement on the GPU): fwd+bwd {timed(step, steps):.2f} ms/step, forward alone {timed(fwd, steps):.2f} ms")
else:
    mod = ax.CrossClipTrackingModule(num_layers=nl, num_classes=K, attn_drop=p, aspp_drop=p, kernel_sizes=[3, 3, 3], atrous_rates=[1, 2, 3],
                                     norm_fn="ln", num_clip_frames=V)
    sd = mod.state_dict()
    sd.update(w)
    mod.load_state_dict(sd, strict=True)
    mod = mod.to(dev).train()
    from axial_vs_amd.cc_training import cc_module_train

    amp_dt = torch.bfloat16 if "--amp" in sys.argv else torch.float16 if "--amp16" in sys.argv else None   # under torch.autocast: 16-bit products

    def step():
        if amp_dt is not None:
            with torch.autocast(device_type="cuda", dtype=amp_dt):
                lg, mk = cc_module_train(mod, cq, pf)
        else:
            lg, mk = cc_module_train(mod, cq, pf)
        torch.autograd.backward([lg, mk], [d_l, d_m])

    def fwd():
        with torch.no_grad():
            cc_module_train(mod, cq, pf)
    print(f"training tier: fwd+bwd {timed(step, steps):.2f} ms/step ({Tc * V / timed(step, steps) * 1e3:.0f} frames/s), forward alone {timed(fwd, steps):.2f} ms")
