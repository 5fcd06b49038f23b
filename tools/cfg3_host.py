"""BASELINE config 3: host-side cost of one forward of the within-clip module (enqueue time without waiting for the GPU, and a
cProfile of where it goes) against the GPU time of the same forward."""
import cProfile, pstats, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import load, weights
from test_cabi_cpu import _decoder_from_meta

z, m = load("g8_pixel_decoder_full_T4_S2")
mod = _decoder_from_meta(dict(m), cross_clip_training=True).eval()
mod.within_clip_tracking_module.load_state_dict(weights(z, m), strict=True)
mod = mod.cuda()
g = torch.Generator().manual_seed(1)
feats = {k: torch.randn(m["B"] * m["T"], m["chans"][k], *m["sizes"][k], generator=g).cuda() for k in m["chans"]}
with torch.no_grad():
    for _ in range(100): mod.forward_features(feats)
    torch.cuda.synchronize()
    for n in (20, 20, 100):
        t0 = time.perf_counter()
        for _ in range(n): mod.forward_features(feats)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"n={n}: enqueue {1e3*(t1-t0)/n:.3f} ms per forward, with the GPU drained {1e3*(t2-t0)/n:.3f} ms")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(200): mod.forward_features(feats)
    pr.disable()
    torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumtime").print_stats(28)
