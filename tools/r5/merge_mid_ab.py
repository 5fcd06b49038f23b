"""Merged q/k/v + trajectory launch on 32-row tiles (T = 5 .. 8), option merge_mid: bit-equality with the two-launch form, error against the float64
oracle, time per layer as a HIP-graph replay:   python3 tools/r5/merge_mid_ab.py [B,T,H,W ...]"""
import sys, os
R = os.path.join(os.path.dirname(__file__), "..", "..")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "oracle"))
import torch
import axvs_oracle as orc
import axial_vs_amd as ax
from axial_vs_amd import _lib
L = _lib.lib()
shapes = [a for a in sys.argv[1:] if "," in a] or ["1,5,24,40", "1,5,32,32", "2,5,24,40", "1,5,25,43", "1,5,40,40", "1,6,32,32", "1,7,25,43", "1,8,24,40", "1,8,32,32", "1,5,48,48", "1,5,64,64", "3,5,24,40"]


def us(g, n=300):
    for _ in range(30): g()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): g()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


def names():
    return [L.axvs_profile_stage_name(i).decode() for i in range(L.axvs_profile_stage_count())]


for sh in shapes:
    B, T, H, W = [int(v) for v in sh.split(",")]
    w = orc.random_weights(orc.axial_layer_param_shapes(256, 1024), 7)
    src, pos = orc.synthetic_clip(B, T, 256, H, W, 7)
    layer = ax.TemporalAxialTrajectoryAttentionLayer(256, 1024, n_heads=8).eval()
    layer.load_state_dict(w, strict=True)
    layer = layer.cuda()
    s = src.cuda()
    p = ax.PositionEmbeddingSine3D(128, normalize=True).channels_last(B, T, H, W, "cuda")
    res = {}
    for mm in (0, 1, 0, 1):
        _lib.check(L.axvs_set_option(b"merge_mid", mm), "axvs_set_option")
        try:
            o = layer(s, p)[0].clone()
            nm = names()
            g = ax.GraphedForward(layer, s, p)
            t = us(g)
            o2 = g()[0].clone()
        finally:
            L.axvs_set_option(b"merge_mid", 0)
        res.setdefault(mm, []).append((t, o, o2, nm))
    # eager outputs only: a GraphedForward replays on CLONES of its inputs, and the clone of a generated position embedding is a plain tensor (read, not generated in the kernel)
    same = torch.equal(res[0][0][1], res[1][0][1]) and torch.equal(res[1][0][1], res[1][1][1]) and torch.equal(res[0][0][2], res[1][0][2])
    err = None
    if B * T * H * W <= 12000:
        ref, _, _ = orc.axial_layer(src.double(), pos.double(), w, 8, want_attn=False)
        err = float((res[1][0][1].cpu().double() - ref).abs().max() / ref.abs().max())
    torch.cuda.synchronize(); ax.check_status()
    print(f"{sh:>12s}: two launches {res[0][0][0]:7.2f} / {res[0][1][0]:7.2f} us   merged {res[1][0][0]:7.2f} / {res[1][1][0]:7.2f} us   "
          f"{'bit-identical' if same else 'DIFFERENT BITS'}   vs float64 {err if err is None else format(err, '.2e')}   {res[1][0][3][1:]}", flush=True)
