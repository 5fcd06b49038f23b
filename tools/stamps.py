"""Phase timeline from a -DAXVS_STAMPS diagnostic build (tools only): python tools/stamps.py <nslots>"""
import sys, os, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
import torch, numpy as np
import axvs_oracle as orc
import axial_vs_amd as ax
from axial_vs_amd import _lib
nslots = int(sys.argv[1]) if len(sys.argv) > 1 else 9
B, T, C, H, W, F = 1, 4, 256, 64, 64, 1024
w = orc.random_weights(orc.axial_layer_param_shapes(C, F), 0)
src, pos = orc.synthetic_clip(B, T, C, H, W, 0)
layer = ax.TemporalAxialTrajectoryAttentionLayer(C, F, n_heads=8).eval()
layer.load_state_dict(w, strict=True)
layer = layer.cuda()
s = src.cuda()
p = pos.cuda() if os.environ.get("AXVS_TENSOR_POS") else ax.PositionEmbeddingSine3D(C // 2, normalize=True).channels_last(B, T, H, W, "cuda")
for _ in range(3): layer(s, p)
torch.cuda.synchronize()
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * (64 * 64))()
raw.axvs_debug_read_stamps(buf, 64 * 64)
a = np.array(buf, dtype=np.uint64).reshape(64, 64).astype(np.int64)
order = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else list(range(nslots))
a = a[order]; nslots = len(order)
d = np.diff(a, axis=0)
print("per-phase cycles (s_memtime ticks = shader cycles... 100MHz const clock on some parts), median over 64 waves:")
for i in range(nslots - 1): print(f"  phase {order[i]}->{order[i+1]}: median {int(np.median(d[i])):8d}  min {int(d[i].min()):8d}  max {int(d[i].max()):8d}")
print("  total", int(np.median(a[nslots-1] - a[0])))
if os.environ.get("AXVS_STAMPS_JSON"):
    import json
    path = os.environ["AXVS_STAMPS_JSON"]
    doc = json.load(open(path)) if os.path.exists(path) else {}
    doc[os.environ.get("AXVS_STAMPS_TAG", "kernel")] = {
        "phases": [{"from": order[i], "to": order[i + 1], "median_cycles": int(np.median(d[i])), "min": int(d[i].min()), "max": int(d[i].max())}
                   for i in range(nslots - 1)],
        "total_median_cycles": int(np.median(a[nslots - 1] - a[0])), "waves": 64,
        "note": "s_memtime stamps (shader cycles) of workgroups 0-7, all 8 waves; -DAXVS_STAMPS diagnostic build"}
    json.dump(doc, open(path, "w"), indent=1)
