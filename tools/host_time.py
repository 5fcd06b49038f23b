"""Host-side cost of enqueueing one layer forward (no GPU wait): python tools/host_time.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
import torch
import axvs_oracle as orc
import axial_vs_amd as ax
from axial_vs_amd import modules as M
B, T, C, H, W, F = 1, 4, 256, 64, 64, 1024
w = orc.random_weights(orc.axial_layer_param_shapes(C, F), 0)
layer = ax.TemporalAxialTrajectoryAttentionLayer(C, F, n_heads=8).eval()
layer.load_state_dict(w, strict=True)
layer = layer.cuda()
src = torch.randn(B * T, H * W, C, device="cuda")
pos = ax.PositionEmbeddingSine3D(C // 2, normalize=True).channels_last(B, T, H, W, "cuda")
for _ in range(5): layer(src, pos)
torch.cuda.synchronize()
for n in (8, 16, 32):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): layer(src, pos)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"enqueue {n} forwards: {(t1 - t0) / n * 1e6:.1f} us each (host only)")
t0 = time.perf_counter()
for _ in range(2000): M._param_key(layer, "f16")
print(f"_param_key: {(time.perf_counter() - t0) / 2000 * 1e6:.1f} us")
t0 = time.perf_counter()
for _ in range(2000): torch.empty_like(src)
print(f"empty_like: {(time.perf_counter() - t0) / 2000 * 1e6:.1f} us")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(200): layer(src, pos)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
