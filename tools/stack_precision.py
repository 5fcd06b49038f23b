"""BASELINE config 3 (full-size within-clip module) against the reference-generated fixture, 16-bit vs fp32 temporal layers:
    python tools/stack_precision.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import load, weights, rel_err, rel_l2, t
from test_cabi_cpu import _decoder_from_meta

for kv in sys.argv[1:]:          # library options: key=value (axvs_set_option)
    from axial_vs_amd import _lib
    k, v = kv.split("=")
    _lib.check(_lib.lib().axvs_set_option(k.encode(), int(v)), k)
z, m = load("g8_pixel_decoder_full_T4_S2")
w = weights(z, m)
g = torch.Generator().manual_seed(m["seed"] + 1)
feats = {k: torch.randn(m["B"] * m["T"], m["chans"][k], *m["sizes"][k], generator=g) for k in m["chans"]}
for prec in ("f16", "f32"):
    mod = _decoder_from_meta(dict(m), cross_clip_training=True).eval()
    mod.within_clip_tracking_module.load_state_dict(w, strict=True)
    mod = mod.cuda().set_stack_precision(prec)
    fc = {k: v.cuda() for k, v in feats.items()}
    with torch.no_grad():
        out, _, _ = mod.forward_features(dict(fc))
        for k in m["chans"]:
            sb = m["sub"][k]
            o = out[k].cpu()[:, ::m["csub"], ::sb, ::sb]
            print(f"{prec} {k}: max/max {rel_err(o, t(z['out_' + k])):.2e} relL2 {rel_l2(o, t(z['out_' + k])):.2e}")
        for _ in range(5):
            mod.forward_features(dict(fc))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            mod.forward_features(dict(fc))
        torch.cuda.synchronize()
        print(f"{prec}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per forward")
