"""Stamps at 7 points of the LAST TWO chunks of the lockstep FFN body (-DAXVS_STAMPS -DAXVS_STAMPS_FFN -DAXVS_STAMPS_FFNC): per-wave deltas."""
import sys, os, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch, numpy as np
import axial_vs_amd as ax
from axial_vs_amd import _lib
layer = ax.TemporalAxialTrajectoryAttentionLayer(256, 1024, n_heads=8).eval().cuda()
s = torch.randn(4, 64 * 64, 256, device="cuda")
p = ax.PositionEmbeddingSine3D(128, normalize=True).channels_last(1, 4, 64, 64, "cuda")
for _ in range(200): layer(s, p)
torch.cuda.synchronize()
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * (64 * 64))()
raw.axvs_debug_read_stamps(buf, 64 * 64)
a = np.array(buf, dtype=np.uint64).reshape(64, 64).astype(np.int64)
names = ["chunk top", "linear1 done", "barrier 1 passed", "activation stored", "barrier 2 passed", "linear2 + sum done"]
for gen, base in (("chunk 2", 32), ("chunk 3", 40)):
    print(gen)
    for i in range(5):
        d = a[base + i + 1] - a[base + i]
        print(f"   {names[i]:22s} -> {names[i + 1]:22s} {int(np.median(d)):6d} (min {int(d.min())}, max {int(d.max())})")
d = a[40] - a[32 + 5]
print(f"   chunk 2 end -> chunk 3 top {int(np.median(d)):6d} (min {int(d.min())}, max {int(d.max())})")
print("FFN half (FSTAMP): ", " ".join(f"{int(np.median(a[i] - a[0])):6d}" for i in range(13)))
# skew: per workgroup, spread of 'linear1 done' over its 8 waves
w = a[41].reshape(8, 8)
print("spread of 'linear1 done' (chunk 3) inside a workgroup: median", int(np.median(w.max(1) - w.min(1))))
