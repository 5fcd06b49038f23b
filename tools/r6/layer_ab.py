"""Layer time (HIP-graph replay and eager) of small-map shapes under sets of library options, with the launch list:
   python tools/r6/layer_ab.py "hsplit=1" "hsplit=0" ["hsplit=1 no_merge_qkv=1" ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import axial_vs_amd as ax
from axial_vs_amd import _lib
L = _lib.lib()
SHAPES = [(1, 2, 25, 43), (1, 2, 24, 40), (1, 5, 12, 20), (1, 5, 24, 40), (1, 4, 16, 16), (1, 4, 32, 32), (1, 2, 49, 85)]
if os.environ.get("SHAPES"):
    SHAPES = [tuple(int(v) for v in s.split(",")) for s in os.environ["SHAPES"].split(";")]
optsets = sys.argv[1:] or ["hsplit=1", "hsplit=0"]
def names():
    return [L.axvs_profile_stage_name(i).decode() for i in range(1, L.axvs_profile_stage_count())]
for (B, T, H, W) in SHAPES:
    layer = ax.TemporalAxialTrajectoryAttentionLayer(256, 1024, n_heads=8).eval().cuda()
    src = torch.randn(B * T, H * W, 256, device="cuda")
    pos = ax.PositionEmbeddingSine3D(128, normalize=True).channels_last(B, T, H, W, "cuda")
    row = []
    for rep in range(2):
        for o in optsets:
            kvs = [kv.split("=") for kv in o.split()]
            for k, v in kvs: _lib.check(L.axvs_set_option(k.encode(), int(v)), k)
            try:
                out = layer(src, pos)[0]; nm = names()
                g = ax.GraphedForward(layer, src, pos)
                for _ in range(30): g()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(200): g()
                e1.record(); torch.cuda.synchronize()
                tg = e0.elapsed_time(e1) / 200 * 1e3
                for _ in range(30): layer(src, pos)
                e0.record()
                for _ in range(200): layer(src, pos)
                e1.record(); torch.cuda.synchronize()
                te = e0.elapsed_time(e1) / 200 * 1e3
                row.append((o, tg, te, nm))
            finally:
                for k, v in kvs: L.axvs_set_option(k.encode(), 0 if k != "hsplit" else 1)
    print(f"[{B},{T},256,{H},{W}]")
    for o, tg, te, nm in row:
        print(f"    {o:28s} graph {tg:7.2f} us   eager {te:7.2f} us   {nm}")
ax.check_status()
