"""Training step (forward + backward) of WithinClipTrackingModule at BASELINE config 3 (ConvNeXt-T pyramid, T = 4, 2 stages x
(1 deformable layer + 2 axial-trajectory layers on res5 / res4), dropout 0.1); --torch: the same math as torch-eager autograd on this
GPU (the oracle decoder in fp32 on the device).   python tools/wc_train_time.py [steps] [--torch]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import axvs_oracle as orc
from golden_util import load, weights
from test_cabi_cpu import _decoder_from_meta

steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 10
z, m = load("g8_pixel_decoder_full_T4_S2")
w = weights(z, m)
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
feats = {k: torch.randn(m["B"] * m["T"], m["chans"][k], *m["sizes"][k], generator=g).to(dev) for k in m["chans"]}
d_out = {k: torch.randn(feats[k].shape, generator=g).to(dev) for k in feats}


def timed(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


if "--torch" in sys.argv:
    wd = {k: v.to(dev).requires_grad_(True) for k, v in w.items()}
    _p2, _p3 = orc.pos_embed_sine_2d, orc.pos_embed_sine_3d
    orc.pos_embed_sine_2d = lambda *a, **k: _p2(*a, **k).to(dev)
    orc.pos_embed_sine_3d = lambda *a, **k: _p3(*a, **k).to(dev)
    _ar = torch.arange
    def step():
        out = orc.pixel_decoder(dict(feats), wd, ["res3", "res4", "res5"], ["res4", "res5"], m["stages"], m["temporal_per_stage"], num_clip_frames=m["T"], B=m["B"])
        sum((out[k] * d_out[k]).sum() for k in out).backward()
    try:
        print(f"torch eager fp32 (oracle decoder on the GPU): fwd+bwd {timed(step, steps):.2f} ms/step")
    except Exception as e:
        print("torch eager path not runnable on the device as written:", str(e)[:200])
else:
    mod = _decoder_from_meta(dict(m), cross_clip_training=True)
    mod.within_clip_tracking_module.load_state_dict(w, strict=True)
    mod = mod.to(dev).train()
    for mm in mod.modules():
        if isinstance(mm, torch.nn.Dropout):
            mm.p = 0.1

    def step():
        out, _, _ = mod.forward_features(dict(feats))
        sum((out[k] * d_out[k]).sum() for k in out).backward()

    def fwd():
        with torch.no_grad():
            mod.forward_features(dict(feats))
    print(f"training tier: fwd+bwd {timed(step, steps):.2f} ms/step ({m['B'] * m['T'] / timed(step, steps) * 1e3:.0f} frames/s), forward alone (train mode, no grad) {timed(fwd, steps):.2f} ms")
