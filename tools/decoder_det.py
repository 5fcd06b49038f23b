import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch
from golden_util import load, weights, t, rel_err
from test_cabi_cpu import _decoder_from_meta
z, m = load("g8_pixel_decoder_T2_S2")
w = weights(z, m)
mod = _decoder_from_meta(m).eval()
mod.within_clip_tracking_module.load_state_dict(w, strict=True)
mod = mod.cuda()
g = torch.Generator().manual_seed(m["seed"] + 1)
feats = {k: torch.randn(m["B"] * m["T"], m["chans"][k], *m["sizes"][k], generator=g) for k in m["chans"]}
outs = []
for i in range(4):
    out, _, _ = mod.forward_features({k: v.cuda() for k, v in feats.items()})
    outs.append({k: v.cpu() for k, v in out.items()})
    print(i, {k: f"{rel_err(outs[-1][k], t(z['out_' + k])):.3e}" for k in out}, {k: bool(torch.equal(outs[-1][k], outs[0][k])) for k in out})
