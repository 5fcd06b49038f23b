#!/usr/bin/env python3
"""Does a kernel's atomicOr reach a word in pinned HOST memory (torch pin_memory = hipHostMalloc), and can the host read it without
a device-to-host copy?  Uses the library's own range-check bit: operands beyond the fp16 range make the fused loaders OR bit 0."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import axial_vs_amd as ax
from axial_vs_amd import _lib
dev = torch.device("cuda", 0)
L = _lib.lib()
w = torch.zeros(4, dtype=torch.int32).pin_memory()
print("pinned ptr", hex(w.data_ptr()))
_lib.check(L.axvs_set_status_buffer(w.data_ptr()), "set_status")
import axial_vs_amd.modules as M
M._status_words.clear()            # keep the wrappers from re-registering a device word
layer = ax.TemporalAxialTrajectoryAttentionLayer(256, 1024, n_heads=8).eval().to(dev)
src = torch.randn(4, 64 * 64, 256, device=dev)
pos = ax.PositionEmbeddingSine3D(128, normalize=True).channels_last(1, 4, 64, 64, dev)
out = layer(src, pos)[0]
torch.cuda.synchronize()
print("in-range run: word =", int(w[0]))
out = layer(src * 1e6, pos)[0]
torch.cuda.synchronize()
print("out-of-range run: word =", int(w[0]), "(expected bit 0 set)")
# timing: does a registered host word slow the normal path?
for name, ptr in (("host word", w.data_ptr()), ("no word", None)):
    _lib.check(L.axvs_set_status_buffer(ptr), "set_status")
    for _ in range(300):
        layer(src, pos)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(500):
        layer(src, pos)
    e1.record()
    torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 500 * 1e3:.2f} us per layer")
