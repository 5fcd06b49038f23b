"""Generate the fixtures of two rarely used reference options (build container only):

* tests/golden/g11_pos3d_mask_*.npz: the REFERENCE PositionEmbeddingSine3D (WC/pos_embeddings.py:68-130) called WITH a padding
  mask (cumulative counts of unmasked positions).  Masks are seeded rectangles of padding at the right / bottom / end of the clip
  plus a few random holes (the cumsum semantics do not assume a rectangle).
* tests/golden/g12_axial_gelu_*.npz: TemporalAxialTrajectoryAttentionLayer(activation="gelu") (WC/temporal_attention.py:9-17).

    python oracle/gen_golden_misc.py
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402


def make_mask(B, T, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    m = torch.zeros(B, T, H, W, dtype=torch.bool)
    for b in range(B):
        ph, pw, pt = (int(torch.randint(0, max(1, d // 3) + 1, (1,), generator=g)) for d in (H, W, T))
        if ph:
            m[b, :, H - ph:] = True
        if pw:
            m[b, :, :, W - pw:] = True
        if pt and T > 1:
            m[b, T - min(pt, T - 1):] = True
    holes = torch.rand(B, T, H, W, generator=g) < 0.03
    return m | holes


def main():
    torch.set_grad_enabled(False)
    ta, pe, _ = gg.load_reference()
    for (B, T, C, H, W, dffn) in [(1, 2, 64, 6, 5, 128), (1, 3, 256, 16, 16, 512)]:
        layer = ta.TemporalAxialTrajectoryAttentionLayer(d_model=C, d_ffn=dffn, n_heads=8, activation="gelu").eval()
        seed = 12000 + T * 100 + H
        shapes, w = gg.load_random(layer, seed)
        g = torch.Generator().manual_seed(seed + 1)
        x = torch.randn(B, T, C, H, W, generator=g)
        src = x.permute(0, 1, 3, 4, 2).reshape(B * T, H * W, C).contiguous()
        pos = pe.PositionEmbeddingSine3D(C // 2, normalize=True)(x, fmt="btchw").permute(0, 1, 3, 4, 2).contiguous()
        out = layer(src, pos)
        out = out[0] if isinstance(out, tuple) else out
        gg.save(f"g12_axial_gelu_B{B}_T{T}_C{C}_H{H}_W{W}", meta=gg.meta(B=B, T=T, C=C, H=H, W=W, d_ffn=dffn, heads=8, seed=seed, shapes=shapes),
                wsum=np.float64(gg.wsum(w)), out=out)
    for (B, T, H, W, n, normalize, scale) in [(2, 3, 6, 7, 16, True, None), (1, 4, 12, 9, 64, True, 3.0), (2, 2, 5, 8, 32, False, None)]:
        seed = 11000 + T * 100 + H * 10 + W
        mask = make_mask(B, T, H, W, seed)
        mod = pe.PositionEmbeddingSine3D(n, normalize=normalize, scale=scale)
        x = torch.zeros(B, T, 2 * n, H, W)
        pos = mod(x, mask=mask, fmt="btchw").permute(0, 1, 3, 4, 2).contiguous()
        gg.save(f"g11_pos3d_mask_B{B}_T{T}_H{H}_W{W}_n{n}", meta=gg.meta(B=B, T=T, H=H, W=W, n=n, normalize=normalize, scale=mod.scale, seed=seed),
                mask=mask.numpy(), pos=pos)


if __name__ == "__main__":
    main()
