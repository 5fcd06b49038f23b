"""Tube-Link trajectory-attention plugin at a realistic size: one clip of T = 5 frames, 3 levels (1/8, 1/16, 1/32 of a
360 x 640 frame: 45 x 80, 23 x 40, 12 x 20), temporal encoder (1 axial layer) on the two coarsest levels.  ms per forward."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import axial_vs_amd as ax

T, shapes, tl = 5, [(12, 20), (23, 40), (45, 80)], 2      # coarse -> fine like the TL decoder's level order
mod = ax.MultiScaleDeformableAxialTrajectoryAttention(embed_dims=256, num_heads=8, num_levels=3, num_temporal_levels=tl, num_temporal_layers=1,
                                                      num_temporal_dim=1024, num_points=4, dropout=0.0, batch_first=False).eval().cuda()
with torch.no_grad():
    for p in mod.parameters():
        if p.dim() > 1: torch.nn.init.xavier_uniform_(p)
    mod.gamma.fill_(0.5) if hasattr(mod, "gamma") else None
nq = sum(h * w for h, w in shapes)
g = torch.Generator(device="cuda").manual_seed(0)
q = torch.randn(nq, T, 256, device="cuda", generator=g)
qp = torch.randn(nq, T, 256, device="cuda", generator=g) * 0.5
pe = ax.PositionEmbeddingSine3D(128, normalize=True)
pos3d = [pe.channels_last(1, T, h, w, "cuda") for (h, w) in shapes[:tl]]
refs = []
for (h, w) in shapes:
    ys, xs = torch.meshgrid((torch.arange(h, device="cuda") + 0.5) / h, (torch.arange(w, device="cuda") + 0.5) / w, indexing="ij")
    refs.append(torch.stack([xs.reshape(-1), ys.reshape(-1)], -1))
ref = torch.cat(refs, 0)[None, :, None].repeat(T, 1, 3, 1).contiguous()
ss = torch.as_tensor(shapes, dtype=torch.long, device="cuda")
run = lambda: mod(query=q, query_pos=qp, query_pos3d=pos3d, key_padding_mask=None, reference_points=ref, spatial_shapes=ss)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
with torch.no_grad():
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2: run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): run()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"TL plugin, T={T}, levels {shapes} ({nq} queries per frame): {dt*1e3:.3f} ms per forward ({T/dt:.0f} frames/s)")
