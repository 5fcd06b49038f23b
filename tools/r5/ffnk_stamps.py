"""Per-k-step stamps of ONE chunk of the lockstep FFN body (-DAXVS_STAMPS -DAXVS_STAMPS_FFN -DAXVS_STAMPS_FFNK=<chunk>): cycles since the chunk
began, median over 8 workgroups x 8 waves.   AXVS_LIB_PATH=tools/ab/<diag>.so python3 tools/r5/ffnk_stamps.py"""
import sys, os, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch, numpy as np
import axial_vs_amd as ax
from axial_vs_amd import _lib
B, T, C, H, W, F = 1, 4, 256, 64, 64, 1024
layer = ax.TemporalAxialTrajectoryAttentionLayer(C, F, n_heads=8).eval().cuda()
s = torch.randn(B * T, H * W, C, device="cuda")
p = ax.PositionEmbeddingSine3D(C // 2, normalize=True).channels_last(B, T, H, W, "cuda")
for _ in range(200): layer(s, p)
torch.cuda.synchronize()
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * (64 * 64))()
raw.axvs_debug_read_stamps(buf, 64 * 64)
a = np.array(buf, dtype=np.uint64).reshape(64, 64).astype(np.int64)
K = int(os.environ.get("FFNK", "2"))
prev_end = {1: 6, 2: 7, 3: 8}.get(K)          # stamp at the end of chunk K - 1 (chunks 0, 1, 2 end at stamps 6, 7, 8)
this_end = {0: 6, 1: 7, 2: 8}.get(K)
seq = ([("end of previous chunk", prev_end)] if prev_end is not None else []) + [("chunk top", 14)] + [(f"linear1 k{j}", 16 + j) for j in range(8)] + \
      [("linear2 start (after activation + barriers)", 13)] + [(f"linear2 k{j}", 24 + j) for j in range(8)] + ([("chunk end", this_end)] if this_end is not None else [])
print("per-wave deltas (median / min / max over 64 waves):")
for (n0, s0), (n1, s1) in zip(seq[:-1], seq[1:]):
    d = a[s1] - a[s0]
    print(f"  {n0:46s} -> {n1:46s} {int(np.median(d)):6d} {int(d.min()):6d} {int(d.max()):6d}")
t0 = a[14]                      # chunk start
med = lambda x: int(np.median(x))
print("FFN half: ", " ".join(f"{med(a[i] - a[0]):6d}" for i in range(13)))
print("chunk start (since FFN start)", med(t0 - a[0]))
print("linear1 k-steps:", " ".join(f"{med(a[16 + j] - t0):6d}" for j in range(8)))
print("linear2 start   :", med(a[13] - t0))
print("linear2 k-steps:", " ".join(f"{med(a[24 + j] - t0):6d}" for j in range(8)))
d1 = [med(a[16 + j] - (a[16 + j - 1] if j else t0)) for j in range(8)]
d2 = [med(a[24 + j] - (a[24 + j - 1] if j else a[13])) for j in range(8)]
print("linear1 per k-step:", d1)
print("linear2 per k-step:", d2)
