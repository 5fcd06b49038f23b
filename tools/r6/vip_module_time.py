"""The within-clip module at the shipped VIPSeg ResNet-50 setting (769 x 1345, T = 2): wall clock, GPU time and host enqueue time per forward, eager and as a
HIP-graph replay.   python3 tools/r6/vip_module_time.py [repeats]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
import axial_vs_amd as ax
from axial_vs_amd import _lib


class _Shape:
    def __init__(self, c, s_):
        self.channels, self.stride = c, s_


dev = torch.device("cuda:0")
chv, szv = {"res3": 512, "res4": 1024, "res5": 2048}, {"res3": (97, 169), "res4": (49, 85), "res5": (25, 43)}
wv = ax.WithinClipTrackingModule(
    {k: _Shape(c, st_) for (k, c), st_ in zip(chv.items(), (8, 16, 32))}, transformer_dropout=0.0, transformer_attn_drop=0.0,
    transformer_nheads=8, transformer_dim_feedforward=1024, transformer_num_stages=2, transformer_spatial_layers=2,
    transformer_temporal_layers=4, transformer_temporal_attn_type="axial-trajectory", transformer_conv_dims=256,
    transformer_spatial_in_features=["res3", "res4", "res5"], transformer_temporal_in_features=["res4", "res5"],
    num_clip_frames=2, cross_clip_training=True).eval().to(dev)
g = torch.Generator(device=dev).manual_seed(1)
feats = {k: torch.randn(2, chv[k], *szv[k], device=dev, generator=g) for k in chv}
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for _ in range(reps):
    with torch.no_grad():
        for _ in range(5):
            wv.forward_features(dict(feats))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record()
        for _ in range(20):
            wv.forward_features(dict(feats))
        e1.record(); torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 20
        t1 = time.perf_counter()
        for _ in range(20):
            wv.forward_features(dict(feats))
        host = (time.perf_counter() - t1) / 20
        torch.cuda.synchronize()
    print(f"eager: wall {wall * 1e3:.3f} ms, events {e0.elapsed_time(e1) / 20:.3f} ms per forward; host enqueue {host * 1e3:.3f} ms", flush=True)
