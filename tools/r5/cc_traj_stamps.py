"""Phase timeline of the 16-row trajectory kernel at the cross-clip shape (BASELINE config 4: one sequence of 4 clips x 128 queries), from a
-DAXVS_STAMPS build:   AXVS_LIB_PATH=tools/diag_stamps.so python3 tools/r5/cc_traj_stamps.py"""
import sys, os, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch, numpy as np
import axial_vs_amd as ax
from axial_vs_amd import _lib
dev = torch.device("cuda:0")
Q, Tc, V, Hc, Wc = 128, 4, 4, 64, 64
cc = ax.CrossClipTrackingModule(num_layers=1, num_classes=124, attn_drop=0.0, aspp_drop=0.0, kernel_sizes=[3, 3, 3], atrous_rates=[1, 2, 3],
                                norm_fn="ln", num_clip_frames=V).eval().to(dev)
cc.eval_outputs_on_cpu = False
g = torch.Generator(device=dev).manual_seed(4)
cq = torch.randn(1, Q, Tc, 256, device=dev, generator=g)
pf = torch.nn.functional.normalize(torch.randn(1, 128, Tc * V, Hc, Wc, device=dev, generator=g), dim=1)
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * (64 * 64))()
order = [0, 11, 12, 13, 14, 15, 1, 2, 3, 4, 5, 9, 6, 7, 8, 10]
tot, ph = [], []
for rep in range(20):
    for _ in range(3): cc(cq, pf)
    torch.cuda.synchronize()
    raw.axvs_debug_read_stamps_mt1(buf, 64 * 64)
    a = np.array(buf, dtype=np.uint64).reshape(64, 64).astype(np.int64)[order]
    ph.append(np.median(np.diff(a, axis=0), axis=1)); tot.append(np.median(a[-1] - a[0]))
ph = np.median(np.array(ph), axis=0)
names = {(0, 11): "start -> q fragments + first K requested", (11, 12): "frame 0 (scores, softmax, AV, x store)", (12, 13): "frame 1", (13, 14): "frame 2", (14, 15): "frame 3",
         (15, 1): "barrier after the spatial half"}
for i in range(len(order) - 1):
    print(f"  {order[i]:2d} -> {order[i + 1]:2d}: {int(ph[i]):7d} cycles   {names.get((order[i], order[i + 1]), '')}")
print("  total", int(np.median(tot)), "cycles (median over 64 waves of workgroups 0-7, 20 forwards)")
