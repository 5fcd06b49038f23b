"""Merged q/k/v + trajectory launch on 16-row tiles (option merge_small; T <= 4, at most 64 tiles of 64 rows): time per layer as a HIP-graph replay, off / on, two rounds."""
import sys, os
R = os.path.join(os.path.dirname(__file__), "..", "..")
sys.path.insert(0, R)
import torch
import axial_vs_amd as ax
from axial_vs_amd import _lib
L = _lib.lib()
shapes = [a for a in sys.argv[1:] if "," in a] or ["1,4,8,8", "1,4,16,16", "1,4,16,24", "1,4,16,32", "1,4,24,24", "1,4,24,32", "1,4,32,32", "1,2,25,43", "1,2,24,40", "1,2,32,32", "1,2,16,16",
                                                    "1,3,16,16", "1,3,24,40", "1,1,32,32", "2,4,16,16", "2,2,25,43", "1,4,12,20", "1,4,24,40"]
for sh in shapes:
    B, T, H, W = [int(v) for v in sh.split(",")]
    layer = ax.TemporalAxialTrajectoryAttentionLayer(256, 1024, n_heads=8).eval().cuda()
    s = torch.randn(B * T, H * W, 256, device="cuda")
    p = ax.PositionEmbeddingSine3D(128, normalize=True).channels_last(B, T, H, W, "cuda")
    res, outs, nm = {0: [], 1: []}, {}, {}
    for mm in (0, 1, 0, 1):
        _lib.check(L.axvs_set_option(b"merge_small", mm), "axvs_set_option")
        try:
            outs[mm] = layer(s, p)[0].clone()
            nm[mm] = [L.axvs_profile_stage_name(i).decode() for i in range(1, L.axvs_profile_stage_count())]
            g = ax.GraphedForward(layer, s, p)
            for _ in range(30): g()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(300): g()
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 300 * 1e3)
            res[mm].append(best)
        finally:
            L.axvs_set_option(b"merge_small", 0)
    Hp, Wp = (H + 15) // 16 * 16, (W + 15) // 16 * 16
    th, tw = B * W * ((T * Hp + 15) // 16), B * H * ((T * Wp + 15) // 16)
    print(f"{sh:>10s} tiles16 {th:4d}/{tw:4d}: two launches {res[0][0]:6.2f} / {res[0][1]:6.2f}   merged {res[1][0]:6.2f} / {res[1][1]:6.2f}   "
          f"{'bit-identical' if torch.equal(outs[0], outs[1]) else 'DIFFERENT BITS'}  {nm[1]}", flush=True)
