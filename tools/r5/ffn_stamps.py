"""FFN-internal phase stamps of the width-pass kernel (-DAXVS_STAMPS -DAXVS_STAMPS_FFN builds): cycles since the FFN half began, median over
8 workgroups, separately for the linear1 waves (0-3) and the linear2 waves (4-7) of the wave-specialised body.
    AXVS_LIB_PATH=tools/ab/<diag>.so python3 tools/r5/ffn_stamps.py [nslots]"""
import sys, os, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch, numpy as np
import axial_vs_amd as ax
from axial_vs_amd import _lib
nslots = int(sys.argv[1]) if len(sys.argv) > 1 else 13
B, T, C, H, W, F = 1, 4, 256, 64, 64, 1024
layer = ax.TemporalAxialTrajectoryAttentionLayer(C, F, n_heads=8).eval().cuda()
s = torch.randn(B * T, H * W, C, device="cuda")
p = ax.PositionEmbeddingSine3D(C // 2, normalize=True).channels_last(B, T, H, W, "cuda")
for _ in range(200): layer(s, p)
torch.cuda.synchronize()
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * (64 * 64))()
raw.axvs_debug_read_stamps(buf, 64 * 64)
a = np.array(buf, dtype=np.uint64).reshape(64, 8, 8).astype(np.int64)[:nslots]      # [slot][workgroup][wave]
rel = a - a[0:1]
for name, ws in (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8))):
    print(name, " ".join(f"{int(np.median(rel[i][:, ws])):6d}" for i in range(nslots)))
print("per-wave end  ", " ".join(f"{int(np.median(rel[nslots - 1][:, w])):6d}" for w in range(8)))
