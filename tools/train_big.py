import os, sys, time, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import axvs_oracle as orc
import axial_vs_amd as ax
B, T, C, H, W, F = 8, 4, 256, 96, 96, 1024
layer = ax.TemporalAxialTrajectoryAttentionLayer(C, F, dropout=0.1, attn_drop=0.1, n_heads=8)
layer.load_state_dict(orc.random_weights(orc.axial_layer_param_shapes(C, F), 1), strict=True)
layer = layer.cuda().train()
g = torch.Generator(device="cuda").manual_seed(0)
s = torch.randn(B * T, H * W, C, device="cuda", generator=g).requires_grad_(True)
p = ax.PositionEmbeddingSine3D(C // 2, normalize=True).channels_last(B, T, H, W, "cuda")
go = torch.randn(B * T, H * W, C, device="cuda", generator=g)
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = layer(s, p)[0]; out.backward(go)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"step {i}: {dt*1e3:.1f} ms, out finite {bool(torch.isfinite(out).all())}, grad finite {bool(torch.isfinite(s.grad).all())}, "
          f"|dW| {float(layer.linear1.weight.grad.norm()):.3e}, peak mem {torch.cuda.max_memory_allocated()/1e9:.1f} GB")
