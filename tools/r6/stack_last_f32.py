"""BASELINE config 3 at full size against the reference fixture: max-norm / relative L2 of the free-running stack and ms per forward with the temporal layers on
16-bit operands, with the LAST temporal layer of every stage on the fp32 tier, and with all of them on it."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from golden_util import load, weights, rel_err, rel_l2, t
from test_hip_parity import _full_size_decoder
z, m = load("g8_pixel_decoder_full_T4_S2")
g = torch.Generator().manual_seed(m["seed"] + 1)
feats = {k: torch.randn(m["B"] * m["T"], m["chans"][k], *m["sizes"][k], generator=g).cuda() for k in m["chans"]}
for prec in ("f16", "f16+final_f32", "f16+last_f32", "f32", "f16", "f16+final_f32"):
    mod = _full_size_decoder(m, weights(z, m)).set_stack_precision(prec)
    with torch.no_grad():
        out, _, _ = mod.forward_features(dict(feats))
        errs = []
        for k in m["chans"]:
            sb = m["sub"][k]
            o = out[k].cpu()[:, ::m["csub"], ::sb, ::sb]
            errs.append(f"{k} {rel_err(o, t(z['out_' + k])):.2e} / {rel_l2(o, t(z['out_' + k])):.2e}")
        for _ in range(10): mod.forward_features(dict(feats))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): mod.forward_features(dict(feats))
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
    print(f"{prec:14s} {dt * 1e3:6.3f} ms per forward   max-norm / relL2: " + "   ".join(errs))
