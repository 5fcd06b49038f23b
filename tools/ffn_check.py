"""Quick GPU check of the metric-shape golden fixture (tuning helper)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
import torch
import __graft_entry__ as ge
ge.build()
import axial_vs_amd as ax
from golden_util import load, weights, axial_inputs, rel_err, t
for name in sys.argv[1:] or ["g2_axial_B1_T4_C256_H64_W64"]:
    z, m = load(name)
    w = weights(z, m)
    src, pos = axial_inputs(m)
    layer = ax.TemporalAxialTrajectoryAttentionLayer(m["C"], m["d_ffn"], n_heads=8).eval()
    layer.load_state_dict(w, strict=True)
    layer = layer.cuda()
    from axial_vs_amd import _lib
    for g in (1, 0):
        _lib.lib().axvs_set_option(b"generic_only", g)
        errs = []
        for it in range(8):
            out = layer(src.cuda(), pos.cuda())[0].cpu()
            errs.append(rel_err(out[:, ::m["stride"]], t(z["out"])))
        print(name, "generic" if g else "fused", " ".join(f"{e:.2e}" for e in errs))
