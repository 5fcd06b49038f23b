import sys, os, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import axvs_oracle as orc, axial_vs_amd as ax
C, F = 256, 1024
layer = ax.TemporalAxialTrajectoryAttentionLayer(C, F, n_heads=8).eval()
layer.load_state_dict(orc.random_weights(orc.axial_layer_param_shapes(C, F), 5), strict=True)
layer = layer.cuda()
pg = ax.PositionEmbeddingSine3D(C // 2, normalize=True).channels_last(1, 4, 64, 64, "cuda")
src = torch.randn(4, 4096, C, device="cuda")
for _ in range(50): layer(src, pg)
torch.cuda.synchronize()
# host-only cost: enqueue while the GPU is kept busy by a long kernel queue? simply measure enqueue of 2000 forwards and total
t0 = time.perf_counter()
for _ in range(2000): layer(src, pg)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e6*(t1-t0)/2000:.1f} us per forward (host), total {1e6*(t2-t0)/2000:.1f} us per forward")
