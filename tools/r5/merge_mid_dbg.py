import sys, os
R = os.path.join(os.path.dirname(__file__), "..", "..")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "oracle"))
import torch
import axvs_oracle as orc
import axial_vs_amd as ax
from axial_vs_amd import _lib
L = _lib.lib()
for sh in ["1,5,32,32", "1,5,24,40"]:
    B, T, H, W = [int(v) for v in sh.split(",")]
    w = orc.random_weights(orc.axial_layer_param_shapes(256, 1024), 7)
    src, pos = orc.synthetic_clip(B, T, 256, H, W, 7)
    layer = ax.TemporalAxialTrajectoryAttentionLayer(256, 1024, n_heads=8).eval()
    layer.load_state_dict(w, strict=True)
    layer = layer.cuda()
    s = src.cuda()
    p = ax.PositionEmbeddingSine3D(128, normalize=True).channels_last(B, T, H, W, "cuda")
    outs = {}
    for mm in (0, 1):
        L.axvs_set_option(b"merge_mid", mm)
        outs[mm] = [layer(s, p)[0].clone() for _ in range(4)]
        L.axvs_set_option(b"merge_mid", 0)
    torch.cuda.synchronize()
    print(sh, "two-launch repeatable:", all(torch.equal(outs[0][0], o) for o in outs[0]), "merged repeatable:", all(torch.equal(outs[1][0], o) for o in outs[1]))
    d = (outs[0][0] - outs[1][0]).abs()
    print("  max |two - merged|", float(d.max()), "rows differing", int((d.amax(-1) > 0).sum()), "of", d.shape[0] * d.shape[1], " max |out|", float(outs[0][0].abs().max()))
    bad = (d.amax(-1) > 0).nonzero()
    print("  first differing (frame, token):", bad[:8].tolist())
    # teacher-forced halves: trajectory attention alone (pass outputs) via the stage options is not exposed; compare with tensor positions instead
    L.axvs_set_option(b"merge_mid", 1)
    o_t = layer(s, pos.cuda())[0].clone()
    L.axvs_set_option(b"merge_mid", 0)
    o_t0 = layer(s, pos.cuda())[0].clone()
    print("  with pos as a tensor: equal", torch.equal(o_t, o_t0), float((o_t - o_t0).abs().max()))
    for mm in (0, 1):
        L.axvs_set_option(b"merge_mid", mm)
        e = layer(s, p)[0].clone()
        g = ax.GraphedForward(layer, s, p)
        o1 = g()[0].clone()
        for _ in range(200): g()
        o2 = g()[0].clone()
        L.axvs_set_option(b"merge_mid", 0)
        torch.cuda.synchronize()
        print(f"  merge_mid {mm}: graph == eager {torch.equal(e, o1)} ({float((e - o1).abs().max())}); after 200 replays {torch.equal(e, o2)} ({float((e - o2).abs().max())})")
