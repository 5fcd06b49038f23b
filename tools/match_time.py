import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import axial_vs_amd as ax
from scipy.optimize import linear_sum_assignment
for Q in (100, 128):
    g = torch.Generator().manual_seed(0)
    tgt = torch.randn(Q, 256, generator=g); cur = tgt[torch.randperm(Q, generator=g)] + 0.3 * torch.randn(Q, 256, generator=g)
    t, c = tgt.cuda(), cur.cuda()
    for _ in range(3): ax.match_from_embds(t, c)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): ax.match_from_embds(t, c)
    e1.record(); torch.cuda.synchronize()
    dev_us = e0.elapsed_time(e1) / 50 * 1e3
    t0 = time.perf_counter()
    for _ in range(50):
        cn = c / c.norm(dim=1)[:, None]; tn = t / t.norm(dim=1)[:, None]
        Cm = (1 - torch.mm(cn, tn.transpose(0, 1))).cpu()
        linear_sum_assignment(Cm.transpose(0, 1))
    ref_us = (time.perf_counter() - t0) / 50 * 1e6
    print(f"Q={Q}: device {dev_us:.0f} us (no sync)   reference path (GPU mm + .cpu() + SciPy) {ref_us:.0f} us")
# the whole clip-alignment loop of a batch of videos: one library call (normalise, pair costs, chained assignment)
B, Tc, Q = 2, 8, 128
g = torch.Generator().manual_seed(1)
base = torch.randn(B, 1, Q, 256, generator=g)
emb = torch.stack([torch.stack([base[b, 0][torch.randperm(Q, generator=g)] + 0.3 * torch.randn(Q, 256, generator=g) for _ in range(Tc)]) for b in range(B)]).cuda()
cen = torch.randn(B, Tc, Q, 256, generator=g).cuda()
for _ in range(3): ax.match_clips(emb, cen)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ax.match_clips(emb, cen)
e1.record(); torch.cuda.synchronize()
print(f"match_clips B={B} Tc={Tc} Q={Q}: {e0.elapsed_time(e1) / 20 * 1e3:.0f} us for {B * (Tc - 1)} assignments "
      f"({e0.elapsed_time(e1) / 20 * 1e3 / (Tc - 1):.0f} us per clip step, {B} videos side by side)")
