"""Generate tests/golden/g10_train_*.npz: the REFERENCE layer in train() mode under autograd.

TemporalAxialTrajectoryAttentionLayer (WC/temporal_attention.py:158-220) is run in float64 with every nn.Dropout replaced by a
module that multiplies by `axvs_oracle.dropout_keep(seed, site, ...)` -- the hash-generated factors the training tier uses --
so the fixture pins WHERE the reference applies dropout (attention maps :55, dropout1 twice :204/:213, dropout2/3 :182-183) and
the gradients that follow, not torch's RNG stream.  Saved: the forward output and the gradients of src, pos and every parameter
for a seeded upstream gradient.  Runs only in the build container.

    python oracle/gen_golden_train.py
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import axvs_oracle as orc  # noqa: E402
import gen_golden as gg  # noqa: E402


class HashDropout(torch.nn.Module):
    """Stand-in for nn.Dropout(p): call i multiplies by the factors of site sites[i]."""

    def __init__(self, p, seed, sites):
        super().__init__()
        self.p, self.seed, self.sites, self.calls = p, seed, list(sites), 0

    def forward(self, x):
        site = self.sites[self.calls]
        self.calls += 1
        return x * orc.dropout_keep(self.seed, site, x.numel(), self.p, x.dtype).reshape(x.shape)


def main():
    torch.manual_seed(0)
    ta, pe, _ = gg.load_reference()
    for (B, T, C, H, W, dffn, p_drop, p_attn) in [(1, 2, 64, 5, 6, 128, 0.0, 0.0), (2, 3, 64, 4, 5, 128, 0.1, 0.2),
                                                 (1, 2, 256, 8, 8, 256, 0.1, 0.1), (1, 4, 128, 6, 3, 256, 0.25, 0.0)]:
        layer = ta.TemporalAxialTrajectoryAttentionLayer(d_model=C, d_ffn=dffn, dropout=p_drop, attn_drop=p_attn, n_heads=8)
        seed = 10000 + T * 100 + H * 10 + W
        shapes, w = gg.load_random(layer, seed)
        layer = layer.double().train()
        dseed = seed * 7 + 1
        layer.height_attn.attn_drop = HashDropout(p_drop, dseed, [1])
        layer.width_attn.attn_drop = HashDropout(p_drop, dseed, [3])
        layer.dropout1 = HashDropout(p_attn, dseed, [2, 4])
        layer.dropout2 = HashDropout(p_drop, dseed, [5])
        layer.dropout3 = HashDropout(p_drop, dseed, [6])
        g = torch.Generator().manual_seed(seed + 1)
        x = torch.randn(B, T, C, H, W, generator=g)
        src = x.permute(0, 1, 3, 4, 2).reshape(B * T, H * W, C).contiguous().double().requires_grad_(True)
        pos = pe.PositionEmbeddingSine3D(C // 2, normalize=True)(x, fmt="btchw").permute(0, 1, 3, 4, 2).contiguous().double().requires_grad_(True)
        d_out = torch.randn(B * T, H * W, C, generator=g).double()
        out = layer(src, pos)
        out = out[0] if isinstance(out, tuple) else out
        out.backward(d_out)
        arrs = dict(meta=gg.meta(B=B, T=T, C=C, H=H, W=W, d_ffn=dffn, heads=8, seed=seed, dropout_seed=dseed, p_dropout=p_drop,
                                 p_attn_drop=p_attn, shapes=shapes),
                    wsum=np.float64(gg.wsum(w)), out=out.detach().float(), d_src=src.grad.float(), d_pos=pos.grad.float())
        for k, v in layer.named_parameters():      # big weight gradients: every 5th element + (sum, sum of squares, max |.|)
            gr = v.grad
            arrs["grad." + k] = gr.float() if gr.numel() <= 20000 else gr.reshape(-1)[::5].float()
            arrs["gradchk." + k] = gg.checks(gr)
        gg.save(f"g10_train_B{B}_T{T}_C{C}_H{H}_W{W}", **arrs)


if __name__ == "__main__":
    main()
