"""Training step (forward + backward) of one layer at BASELINE config 2 through the training tier: ms per step, frames/s."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import axvs_oracle as orc
import axial_vs_amd as ax

B, T, C, H, W, F = 1, 4, 256, 64, 64, 1024
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
from axial_vs_amd import _lib
for a in sys.argv[2:]:          # library options: key=value
    if "=" in a:
        k, v = a.split("=")
        _lib.check(_lib.lib().axvs_set_option(k.encode(), int(v)), "axvs_set_option")
        print("option", k, v)
for exact, recompute in ((1, True), (1, False)):
    _lib.lib().axvs_set_option(b"train_exact", exact)
    pd = float(os.environ.get("AXVS_P_DROP", "0.1"))
    layer = ax.TemporalAxialTrajectoryAttentionLayer(C, F, dropout=pd, attn_drop=pd, n_heads=8)
    layer.load_state_dict(orc.random_weights(orc.axial_layer_param_shapes(C, F), 1), strict=True)
    layer = layer.cuda().train()
    layer.recompute = recompute
    src, pos = orc.synthetic_clip(B, T, C, H, W, 1)
    s, p = src.cuda().requires_grad_(True), pos.cuda()
    g = torch.randn_like(s)
    amp = {"--amp": torch.bfloat16, "--amp16": torch.float16}
    amp_dt = next((v for k, v in amp.items() if k in sys.argv), None)      # under torch.autocast: 16-bit products in the Linear layers
    def step():
        if amp_dt is not None:
            with torch.autocast(device_type="cuda", dtype=amp_dt):
                out = layer(s, p)[0]
        else:
            out = layer(s, p)[0]
        out.backward(g)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    # forward only
    t0 = time.perf_counter()
    for _ in range(n):
        with torch.no_grad():
            layer(s, p)
    torch.cuda.synchronize()
    df = (time.perf_counter() - t0) / n
    print(f"train_exact={exact} recompute={recompute}: fwd+bwd {dt*1e3:.2f} ms/step ({B*T/dt:.0f} frames/s), forward alone {df*1e3:.2f} ms")

# reference point: the same math as plain PyTorch ops on the same GPU (autograd through the oracle's torch code, fp32, rocBLAS/MIOpen
# kernels chosen by torch) -- what "unmodified reference on ROCm PyTorch" would cost for this layer
if "--torch" in sys.argv:
    orc.dropout_keep = lambda seed, site, count, p, dtype=torch.float32: torch.ones(count, dtype=dtype, device="cuda")
    w = {k: v.cuda().requires_grad_(True) for k, v in orc.random_weights(orc.axial_layer_param_shapes(C, F), 1).items()}
    src, pos = orc.synthetic_clip(B, T, C, H, W, 1)
    s, p = src.cuda().requires_grad_(True), pos.cuda()
    g = torch.randn_like(s)
    def tstep():
        out = orc.axial_layer_train(s, p, w, 8, 0.0, 0.0, 0)
        out.backward(g)
    for _ in range(3):
        tstep()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        tstep()
    torch.cuda.synchronize()
    print(f"torch eager (oracle code on the GPU, fp32, no dropout): fwd+bwd {(time.perf_counter() - t0) / n * 1e3:.2f} ms/step")
    with torch.autocast("cuda", dtype=torch.float16):
        for _ in range(2):
            tstep()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            tstep()
        torch.cuda.synchronize()
    print(f"torch eager under fp16 autocast: fwd+bwd {(time.perf_counter() - t0) / n * 1e3:.2f} ms/step")
