"""One-off wide sweep (not part of the test suite): random shapes of the axial layer against the float64 oracle."""
import os, sys, random, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import axvs_oracle as orc
import axial_vs_amd as ax

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
worst, t0 = (0.0, None), time.time()
for i in range(n):
    C = rng.choice([64, 128, 256, 256, 256])
    T = rng.randint(1, 8)
    big = rng.random() < 0.5
    H, W = (rng.randint(1, 130), rng.randint(1, 130)) if big else (rng.randint(1, 30), rng.randint(1, 30))
    B = rng.randint(1, 3)
    while B * T * H * W > 24000:
        if B > 1: B -= 1
        elif T > 1: T -= 1
        else: H = max(1, H // 2)
    F = rng.choice([128, 256, 512, 1024])
    w = orc.random_weights(orc.axial_layer_param_shapes(C, F), 3000 + i)
    src, pos = orc.synthetic_clip(B, T, C, H, W, 3000 + i)
    ref, _, _ = orc.axial_layer(src.double(), pos.double(), w, 8, want_attn=False)
    layer = ax.TemporalAxialTrajectoryAttentionLayer(C, F, n_heads=8).eval()
    layer.load_state_dict(w, strict=True)
    layer = layer.cuda()
    out, _, _ = layer(src.cuda(), pos.cuda())
    out2, _, _ = layer(src.cuda(), pos.cuda())
    e = float((out.cpu().double() - ref).abs().max() / ref.abs().max())
    same = torch.equal(out, out2)
    flag = "" if (e < 1e-3 and same) else "   <<<<<< FAIL"
    print(f"{i:3d} B{B} T{T} C{C} H{H} W{W} F{F}: {e:.2e} repeat-equal {same}{flag}", flush=True)
    if e > worst[0]:
        worst = (e, (B, T, C, H, W, F))
torch.cuda.synchronize()
ax.check_status()
from axial_vs_amd import modules
print("worst", worst, f"{time.time() - t0:.0f} s;  arrival counters all zero:", all(int(b.abs().sum()) == 0 for b in modules._sync_buffers.values()))
