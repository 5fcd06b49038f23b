"""BASELINE config 3: WithinClipTrackingModule.forward_features at ConvNeXt-T size (res3 [4,192,64,64], res4 [4,384,32,32],
res5 [4,768,16,16], T = 4, 2 stages x (1 deformable spatial layer + 2 axial-trajectory layers on res5 / res4)): ms per forward."""
import os, sys, time
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
from axial_vs_amd import _lib
for a in sys.argv[2:]:
    if "=" in a:
        k, v = a.split("=")
        if k == "inplace":                    # inplace=0: the temporal levels are split out of / written back into the token buffer
            from axial_vs_amd import pixel_decoder as _pd
            _pd._IN_PLACE_LEVELS = bool(int(v))
            print("in-place levels", bool(int(v)))
            continue
        _lib.check(_lib.lib().axvs_set_option(k.encode(), int(v)), "axvs_set_option")
        print("option", k, v)
with torch.no_grad():
    t_set = time.perf_counter()
    while time.perf_counter() - t_set < 0.3:
        mod.forward_features(feats)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        mod.forward_features(feats)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
print(f"cfg3 within-clip module forward: {dt*1e3:.3f} ms ({m['B']*m['T']/dt:.0f} frames/s)")

if os.environ.get("AXVS_CFG3_NO_GRAPH", "") not in ("", "0"):
    sys.exit(0)
# the same forward replayed from a captured HIP graph (two streams inside: the temporal levels of a stage run side by side)
keys = list(feats)
def fn(*ts):
    o, _, _ = mod.forward_features({k: t for k, t in zip(keys, ts)})
    return tuple(o[k] for k in keys)
import axial_vs_amd as ax
class _Fn(torch.nn.Module):
    def forward(self, *ts):
        return fn(*ts)
with torch.no_grad():
    gf = ax.GraphedForward(_Fn(), *[feats[k] for k in keys])
    for _ in range(5): gf()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        gf()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    eager = fn(*[feats[k] for k in keys])
    same = all(torch.equal(a, b) for a, b in zip(eager, gf()))
print(f"cfg3 within-clip module forward, HIP graph replay: {dt*1e3:.3f} ms ({m['B']*m['T']/dt:.0f} frames/s), bit-equal to eager: {same}")
