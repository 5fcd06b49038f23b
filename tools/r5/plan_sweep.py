"""Where do the few-rows forms (16-row trajectory tiles + stand-alone FFN) stop paying?  One layer at shapes between 16 and 256 tiles of
64 rows, timed with option small_tiles_below = 1 (always the 64-row / merged / FFN-riding forms), 128 (the threshold until round 5; 65 since) and 1 << 20 (always the
few-rows forms):    python3 tools/r5/plan_sweep.py [B,T,H,W ...]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
import axial_vs_amd as ax
from axial_vs_amd import _lib

DEFAULT = ["1,4,16,16", "1,4,32,32", "1,4,32,40", "1,4,32,48", "1,4,40,40", "1,4,32,64", "1,4,48,48", "1,4,64,64",
           "1,2,24,40", "1,2,25,43", "1,2,32,32", "1,2,32,64", "1,2,48,48", "1,2,48,64", "1,2,48,80", "1,2,64,64", "1,2,49,85",
           "1,3,32,32", "1,3,40,40", "1,3,48,48", "1,1,64,64", "1,1,96,96", "2,4,16,32", "2,2,32,32"]
shapes = [a for a in sys.argv[1:] if "," in a] or DEFAULT


def us(layer, s, p, n=300):
    for _ in range(30): layer(s, p)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): layer(s, p)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


print(f"{'shape':>14s} {'rows':>6s} {'tiles':>5s} | {'64-row forms':>12s} {'threshold 128':>13s} {'few-rows forms':>14s}   (us per layer; graph replay of 1 layer)")
for sh in shapes:
    B, T, H, W = [int(v) for v in sh.split(",")]
    layer = ax.TemporalAxialTrajectoryAttentionLayer(256, 1024, n_heads=8).eval().cuda()
    s = torch.randn(B * T, H * W, 256, device="cuda")
    p = ax.PositionEmbeddingSine3D(128, normalize=True).channels_last(B, T, H, W, "cuda")
    r, outs = [], []
    for thr in (1, 128, 1 << 20):
        _lib.check(_lib.lib().axvs_set_option(b"small_tiles_below", thr), "axvs_set_option")
        try:
            g = ax.GraphedForward(layer, s, p)
            r.append(us(lambda a, b: g(), s, p))
            outs.append(g()[0].clone())
        finally:
            _lib.lib().axvs_set_option(b"small_tiles_below", 0)
    same = all(torch.equal(outs[0], o) for o in outs[1:])
    print(f"{sh:>14s} {B * T * H * W:6d} {B * T * H * W // 64:5d} | {r[0]:12.2f} {r[1]:13.2f} {r[2]:14.2f}   {'bit-identical' if same else 'DIFFERENT BITS'}", flush=True)
