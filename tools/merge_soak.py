"""Soak test of the in-launch K / V^T hand-off of the merged q/k/v kernels under UNEVEN load: a second stream keeps part of the chip busy
with unrelated work of varying size while the layer runs merged and two-launch forwards on fresh inputs; every output word of every
call has to agree.  python tools/merge_soak.py [seconds]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import axvs_oracle as orc  # noqa: E402
import axial_vs_amd as ax  # noqa: E402
from axial_vs_amd import _lib  # noqa: E402

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
C, F = 256, 1024
w = orc.random_weights(orc.axial_layer_param_shapes(C, F), 5)
layer = ax.TemporalAxialTrajectoryAttentionLayer(C, F, n_heads=8).eval()
layer.load_state_dict(w, strict=True)
layer = layer.cuda()
shapes = [(1, 4, 64, 64), (2, 4, 64, 64), (1, 2, 64, 64), (1, 4, 48, 80), (1, 3, 32, 64), (1, 4, 96, 96), (1, 4, 64, 32),
          (1, 2, 49, 85), (1, 4, 49, 85), (2, 3, 57, 61), (1, 4, 33, 70),      # round 5: ragged frames (padded row space) merge too
          (1, 5, 24, 40), (1, 6, 32, 32), (1, 8, 32, 32), (1, 5, 25, 43),     # ... and 5 .. 8 frames per clip on 32-row tiles (one round of the chip)
          (1, 4, 16, 16), (1, 2, 25, 43), (2, 4, 16, 16), (1, 3, 12, 20)]     # ... and few rows on 16-row tiles (passes of up to 128 tiles)
side = torch.cuda.Stream()
noise_a = torch.randn(4096, 4096, device="cuda")
L = _lib.lib()
g = torch.Generator(device="cuda").manual_seed(7)
t0, calls, bad = time.time(), 0, 0
it = 0
while time.time() - t0 < secs:
    B, T, H, W = shapes[it % len(shapes)]
    it += 1
    pg = ax.PositionEmbeddingSine3D(C // 2, normalize=True).channels_last(B, T, H, W, "cuda")
    src = torch.randn(B * T, H * W, C, device="cuda", generator=g)
    _lib.check(L.axvs_set_option(b"no_merge_qkv", 1), "opt")
    ref = layer(src, pg)[0].clone()
    _lib.check(L.axvs_set_option(b"no_merge_qkv", 0), "opt")
    torch.cuda.synchronize()
    outs = []
    for k in range(12):
        with torch.cuda.stream(side):                      # unrelated load of varying size on another stream
            n = 256 * (1 + (it + k) % 16)
            torch.mm(noise_a[:n], noise_a[:, :n])
        outs.append(layer(src, pg)[0])
    torch.cuda.synchronize()
    for o in outs:
        calls += 1
        if not torch.equal(o, ref):
            bad += 1
            print("MISMATCH", (B, T, H, W), float((o - ref).abs().max()))
print(f"{calls} merged forwards under uneven load, {bad} mismatching; stage names of the last: ",
      [L.axvs_profile_stage_name(i).decode() for i in range(L.axvs_profile_stage_count())][1:])
sys.exit(1 if bad else 0)
