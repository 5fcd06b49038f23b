import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import axvs_oracle as orc
import axial_vs_amd as ax
from axial_vs_amd import _lib
from golden_util import rel_err
B, T, C, H, W, F = 1, 4, 256, 96, 60, 256
cnt = {0: 0, 1: 0}
for seed in range(400, 414):
    w = orc.random_weights(orc.axial_layer_param_shapes(C, F), seed)
    src, pos = orc.synthetic_clip(B, T, C, H, W, seed)
    d_out = torch.randn(B * T, H * W, C, generator=torch.Generator().manual_seed(1))
    wd = {k: v.double().requires_grad_(True) for k, v in w.items()}
    sd, pd = src.double().requires_grad_(True), pos.double().requires_grad_(True)
    ref = orc.axial_layer_train(sd, pd, wd, 8, 0.0, 0.0, 1)
    ref.backward(d_out.double())
    line = f"{seed}"
    for opt in (0, 1):
        _lib.lib().axvs_set_option(b"train_attn_split", opt)
        layer = ax.TemporalAxialTrajectoryAttentionLayer(C, F, dropout=0.0, attn_drop=0.0, n_heads=8)
        layer.load_state_dict(w, strict=True)
        layer = layer.cuda().train()
        s = src.cuda().requires_grad_(True); p = pos.cuda().requires_grad_(True)
        out = layer(s, p)[0]
        out.backward(d_out.cuda())
        e = rel_err(s.grad.cpu(), sd.grad.cpu())
        cnt[opt] += e > 1e-4
        line += f"   opt{opt}: out {rel_err(out.detach().cpu(), ref.detach().cpu()):.1e} d_src {e:.1e}"
    print(line, flush=True)
print("runs with a ReLU tie (d_src > 1e-4):", cnt)
