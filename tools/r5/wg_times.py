"""Start / end of every workgroup of the two layer kernels at the metric shape (-DAXVS_STAMPS_WG build; 100 MHz real-time counter, the same on every CU).
    AXVS_LIB_PATH=tools/ab/wg.so python3 tools/r5/wg_times.py"""
import sys, os, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch, numpy as np
import axial_vs_amd as ax
from axial_vs_amd import _lib
shape = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1,4,64,64").split(",")]
B, T, H, W = shape
layer = ax.TemporalAxialTrajectoryAttentionLayer(256, 1024, n_heads=8).eval().cuda()
s = torch.randn(B * T, H * W, 256, device="cuda")
p = ax.PositionEmbeddingSine3D(128, normalize=True).channels_last(B, T, H, W, "cuda")
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * (2 * 1024 * 4))()
res = {0: [], 1: []}
for rep in range(40):
    for _ in range(30): layer(s, p)
    torch.cuda.synchronize()
    raw.axvs_debug_read_wg_times(buf, 2 * 1024 * 4)
    a = np.array(buf, dtype=np.uint64).reshape(2, 1024, 4).astype(np.int64)
    for kind in (0, 1):
        n = min(1024, B * ((T * (H if kind else W) + 63) // 64) * (W if kind else H))
        k = a[kind, :n]
        t0, t1 = k[:, 0].min(), k[:, 1].max()
        dur = (k[:, 1] - k[:, 0]) * 10.0          # ns
        res[kind].append(((t1 - t0) * 10.0, np.median(dur), dur.max(), dur.min(), (k[:, 0].max() - t0) * 10.0, np.median(t1 - k[:, 1]) * 10.0, k))
for kind, name in ((1, "height pass"), (0, "width pass + FFN")):
    r = res[kind]
    med = lambda i: float(np.median([x[i] for x in r]))
    print(f"{name}: launch span {med(0) / 1e3:.2f} us | workgroup duration median {med(1) / 1e3:.2f}, max {med(2) / 1e3:.2f}, min {med(3) / 1e3:.2f} us | "
          f"last workgroup starts {med(4) / 1e3:.2f} us after the first | the median workgroup is done {med(5) / 1e3:.2f} us before the last")
    k = r[-1][6]
    dur = (k[:, 1] - k[:, 0]) * 10.0
    for x in range(8):
        m = k[:, 2] == x
        if m.any(): print(f"    XCC {x}: {int(m.sum()):3d} workgroups, duration median {np.median(dur[m]) / 1e3:.2f} max {dur[m].max() / 1e3:.2f} us, start spread {(k[m, 0].max() - k[:, 0].min()) * 10 / 1e3:.2f} us")
