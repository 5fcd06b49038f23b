"""Python-side cost of one layer forward (cProfile over 3000 calls): where the host time of a step goes."""
import cProfile, pstats, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import axvs_oracle as orc, axial_vs_amd as ax
C, F = 256, 1024
layer = ax.TemporalAxialTrajectoryAttentionLayer(C, F, n_heads=8).eval()
layer.load_state_dict(orc.random_weights(orc.axial_layer_param_shapes(C, F), 5), strict=True)
layer = layer.cuda()
pg = ax.PositionEmbeddingSine3D(C // 2, normalize=True).channels_last(1, 4, 64, 64, "cuda")
src = torch.randn(4, 4096, C, device="cuda")
for _ in range(50): layer(src, pg)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(3000): layer(src, pg)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
