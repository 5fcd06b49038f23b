"""Phase timeline of the 16-row trajectory kernel (width pass: the last writer of the stamps) of one axial layer at a few-rows shape, from a -DAXVS_STAMPS build:
    AXVS_LIB_PATH=tools/diag_stamps.so python3 tools/r5/mt1_stamps.py 1,4,16,16 [1,4,32,32 ...]"""
import sys, os, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch, numpy as np
import axial_vs_amd as ax
from axial_vs_amd import _lib
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * (64 * 64))()
order = [0, 11, 12, 13, 14, 15, 1, 2, 3, 4, 5, 9, 6, 7, 8, 10]
for sh in sys.argv[1:]:
    B, T, H, W = [int(v) for v in sh.split(",")]
    assert T == 4, "the stamp reader belongs to the (f16, T = 4, 16-row) unit"
    layer = ax.TemporalAxialTrajectoryAttentionLayer(256, 1024, n_heads=8).eval().cuda()
    s = torch.randn(B * T, H * W, 256, device="cuda")
    p = ax.PositionEmbeddingSine3D(128, normalize=True).channels_last(B, T, H, W, "cuda")
    ph, tot = [], []
    for rep in range(20):
        for _ in range(3): layer(s, p)
        torch.cuda.synchronize()
        raw.axvs_debug_read_stamps_mt1(buf, 64 * 64)
        a = np.array(buf, dtype=np.uint64).reshape(64, 64).astype(np.int64)[order]
        ph.append(np.median(np.diff(a, axis=0), axis=1)); tot.append(np.median(a[-1] - a[0]))
    ph = np.median(np.array(ph), axis=0)
    print(f"[{sh}] width-pass 16-row kernel: " + " | ".join(f"{order[i]}>{order[i + 1]} {int(ph[i])}" for i in range(len(order) - 1)) + f" | total {int(np.median(tot))}")
