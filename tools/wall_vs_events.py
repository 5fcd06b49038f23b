"""Is the wall-clock step time stable over consecutive timed regions?  python tools/wall_vs_events.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
import torch
import axvs_oracle as orc
import axial_vs_amd as ax
B, T, C, H, W, F = 1, 4, 256, 64, 64, 1024
w = orc.random_weights(orc.axial_layer_param_shapes(C, F), 0)
layer = ax.TemporalAxialTrajectoryAttentionLayer(C, F, n_heads=8).eval()
layer.load_state_dict(w, strict=True)
layer = layer.cuda()
src = torch.randn(B * T, H * W, C, device="cuda")
pos = ax.PositionEmbeddingSine3D(C // 2, normalize=True).channels_last(B, T, H, W, "cuda")
for _ in range(20): layer(src, pos)
torch.cuda.synchronize()
for K in (20, 20, 200, 200, 1000, 200, 20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    for _ in range(K): layer(src, pos)
    e1.record()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(f"K={K}: wall {(t1 - t0) / K * 1e6:.2f} us/step, events {e0.elapsed_time(e1) / K * 1e3:.2f} us/step")
