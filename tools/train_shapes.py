"""Diagnosis helper: training-tier gradients at one shape against float64 autograd on the oracle, with the location of the worst
errors (python tools/train_shapes.py B T C H W F [train_valu])."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, axvs_oracle as orc
import __graft_entry__ as ge
ge.build()
import axial_vs_amd as ax
from axial_vs_amd import _lib
from golden_util import rel_err
B, T, C, H, W, F = [int(x) for x in sys.argv[1:7]]
valu = int(sys.argv[7]) if len(sys.argv) > 7 else 0
recompute = int(sys.argv[8]) if len(sys.argv) > 8 else 1
seed = int(sys.argv[9]) if len(sys.argv) > 9 else 300
exact = int(sys.argv[10]) if len(sys.argv) > 10 else 0
_lib.lib().axvs_set_option(b"train_exact", exact)
pd_, pa_ = (float(sys.argv[11]), float(sys.argv[12])) if len(sys.argv) > 12 else (0.0, 0.0)
dseed = int(sys.argv[13]) if len(sys.argv) > 13 else 1
_lib.lib().axvs_set_option(b"train_valu", valu)
w = orc.random_weights(orc.axial_layer_param_shapes(C, F), seed)
src, pos = orc.synthetic_clip(B, T, C, H, W, seed)
d_out = torch.randn(B * T, H * W, C, generator=torch.Generator().manual_seed(52 if len(sys.argv) > 13 else 0))
wd = {k: v.double().requires_grad_(True) for k, v in w.items()}
sd, pd = src.double().requires_grad_(True), pos.double().requires_grad_(True)
ref = orc.axial_layer_train(sd, pd, wd, 8, pd_, pa_, dseed)
ref.backward(d_out.double())
layer = ax.TemporalAxialTrajectoryAttentionLayer(C, F, dropout=pd_, attn_drop=pa_, n_heads=8)
layer.load_state_dict(w, strict=True); layer = layer.cuda().train(); layer.dropout_seed = dseed; layer.recompute = bool(recompute)
s = src.float().cuda().requires_grad_(True); p = pos.float().cuda().requires_grad_(True)
out, _, _ = layer(s, p); out.backward(d_out.float().cuda())
print((B, T, C, H, W, F), "valu", valu, "recompute", recompute, "seed", seed, "exact", exact, "p", (pd_, pa_), "out %.2e d_src %.2e d_pos %.2e" % (rel_err(out.detach().cpu(), ref.detach()), rel_err(s.grad.cpu(), sd.grad), rel_err(p.grad.cpu(), pd.grad)))
e = (p.grad.cpu().double() - pd.grad).abs().reshape(B, T, H, W, C)
print("d_pos error by t:", e.amax(dim=(0, 2, 3, 4)).tolist())
print("d_pos error by h:", [round(x, 6) for x in e.amax(dim=(0, 1, 3, 4)).tolist()])
print("d_pos error by w:", [round(x, 6) for x in e.amax(dim=(0, 1, 2, 4)).tolist()])
print("d_pos error by head:", [round(x, 6) for x in e.reshape(B, T, H, W, 8, C // 8).amax(dim=(0, 1, 2, 3, 5)).tolist()])
for k, v in layer.named_parameters():
    g = wd[k].grad
    err = float((v.grad.cpu().double() - g).norm() / max(float(g.norm()), 1e-12))
    if err > 1e-4: print("  param", k, "%.2e" % err)
