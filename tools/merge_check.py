"""Merged q/k/v + trajectory launches (axvs_set_sync_buffer) against the two-launch form: bit-identical outputs on fresh inputs
call after call (stale K / V^T of the previous call sit in the L2s and must never be read), then a timing A/B on one box.
    python tools/merge_check.py [B T H W [reps]]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import axvs_oracle as orc  # noqa: E402
import axial_vs_amd as ax  # noqa: E402
from axial_vs_amd import _lib  # noqa: E402


def stage_names():
    L = _lib.lib()
    return [L.axvs_profile_stage_name(i).decode() for i in range(L.axvs_profile_stage_count())]


def opt(name, v):
    _lib.check(_lib.lib().axvs_set_option(name.encode(), v), name)


def run(shape, reps, timing=True):
    B, T, H, W = shape
    C, F = 256, 1024
    w = orc.random_weights(orc.axial_layer_param_shapes(C, F), 5)
    layer = ax.TemporalAxialTrajectoryAttentionLayer(C, F, n_heads=8).eval()
    layer.load_state_dict(w, strict=True)
    layer = layer.cuda()
    pg = ax.PositionEmbeddingSine3D(C // 2, normalize=True).channels_last(B, T, H, W, "cuda")
    ptensor = pg.clone()           # a plain tensor: positions read from HBM
    bad = 0
    for it in range(reps):
        g = torch.Generator(device="cuda").manual_seed(100 + it)
        src = torch.randn(B * T, H * W, C, device="cuda", generator=g)
        for pos in (pg, ptensor):
            opt("no_merge_qkv", 1)
            a = layer(src, pos)[0].clone()
            na = stage_names()
            opt("no_merge_qkv", 0)
            b = layer(src, pos)[0].clone()
            nb = stage_names()
            if it == 0 and pos is pg:
                print(shape, "two-launch:", na[1:], "| merged:", nb[1:])
            if not torch.equal(a, b):
                bad += 1
                d = (a - b).abs()
                print(f"  MISMATCH it={it} gen={pos is pg}: max|d|={float(d.max()):.3e} rows differing={int((d.amax(-1) > 0).sum())}")
    torch.cuda.synchronize()
    print(shape, "bit-identical" if bad == 0 else f"{bad} mismatching calls", f"({2 * reps} comparisons)")
    if not timing:
        return bad
    src = torch.randn(B * T, H * W, C, device="cuda")
    res = {}
    for rnd in range(3):
        for name, v in (("two-launch", 1), ("merged", 0)):
            opt("no_merge_qkv", v)
            for _ in range(30):
                layer(src, pg)
            torch.cuda.synchronize()
            n = 300
            t0 = time.perf_counter()
            for _ in range(n):
                layer(src, pg)
            torch.cuda.synchronize()
            res.setdefault(name, []).append((time.perf_counter() - t0) / n * 1e6)
    opt("no_merge_qkv", 0)
    print(shape, {k: [round(x, 1) for x in v] for k, v in res.items()}, "us per layer")
    return bad


if __name__ == "__main__":
    for kv in [x for x in sys.argv[1:] if "=" in x]:       # library options: key=value
        k, v = kv.split("=")
        opt(k, int(v))
        print("option", k, v)
    a = [int(x) for x in sys.argv[1:] if "=" not in x]
    shapes = [tuple(a[:4])] if len(a) >= 4 else [(1, 4, 64, 64), (2, 4, 64, 64), (1, 2, 64, 64), (1, 3, 32, 64), (1, 4, 48, 80), (1, 4, 96, 96), (1, 1, 64, 64), (3, 4, 16, 32)]
    reps = a[4] if len(a) >= 5 else 6
    bad = 0
    for sh in shapes:
        bad += run(sh, reps)
    sys.exit(1 if bad else 0)
