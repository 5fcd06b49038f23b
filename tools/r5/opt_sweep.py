"""One layer as a HIP-graph replay for a list of values of one integer option:   python3 tools/r5/opt_sweep.py <option> v1,v2,.. B,T,H,W ..."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
import axial_vs_amd as ax
from axial_vs_amd import _lib
L = _lib.lib()
opt, vals = sys.argv[1].encode(), [int(v) for v in sys.argv[2].split(",")]
for sh in sys.argv[3:]:
    B, T, H, W = [int(v) for v in sh.split(",")]
    layer = ax.TemporalAxialTrajectoryAttentionLayer(256, 1024, n_heads=8).eval().cuda()
    s = torch.randn(B * T, H * W, 256, device="cuda")
    p = ax.PositionEmbeddingSine3D(128, normalize=True).channels_last(B, T, H, W, "cuda")
    row, outs = [], []
    for v in vals + vals:
        _lib.check(L.axvs_set_option(opt, v), "axvs_set_option")
        outs.append(layer(s, p)[0].clone())
        nm = [L.axvs_profile_stage_name(i).decode() for i in range(1, L.axvs_profile_stage_count())]
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
        row.append(best)
    L.axvs_set_option(opt, vals[0])
    same = all(torch.equal(outs[0], o) for o in outs)
    print(f"{sh:>11s} ({B * T * H * W // 64:4d} tiles): " + "  ".join(f"{opt.decode()}={v}: {row[i]:.2f} / {row[i + len(vals)]:.2f}" for i, v in enumerate(vals)) + f"   {'same bits' if same else 'DIFFERENT BITS'}  {nm}", flush=True)
