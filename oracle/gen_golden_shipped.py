"""Reference-generated fixtures at the map sizes the SHIPPED configurations run (round 5): the axial-trajectory layer of the reference
(WC/temporal_attention.py:158-220, imported from /root/reference through gen_golden's loader) on
  * VIPSeg ResNet-50 temporal levels, NUM_CLIP_FRAMES 2: [1,2,256,49,85] and [1,2,256,25,43]
    (MaXTron_Video-kMaX/configs/VIPSeg/panoptic_segmentation/maxtron_wc_r50.yaml; WC/msdeformattn.py:248-266),
  * Tube-Link YouTube-VIS 2021 temporal levels, 5 frames: [1,5,256,24,40] and [1,5,256,12,20]
    (MaXTron_Tube-Link/configs/video/ytvis21/ytvis21_r50_maxtron_wc_5k_10k_15k.py).
Frame lengths that are not multiples of 16 (49, 85, 25, 43, 24, 40, 12, 20) are what the padded-frame row space of the HIP kernels exists
for.  Stores the seed, the parameter shape table, a weight checksum, a row-strided subsample of the output and float64 checksums.

    python oracle/gen_golden_shipped.py        # build container only: writes tests/golden/g15_shipped_*.npz
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402

SHAPES = [(1, 2, 256, 49, 85, 1024, 16), (1, 2, 256, 25, 43, 1024, 4), (1, 5, 256, 24, 40, 1024, 8), (1, 5, 256, 12, 20, 1024, 2)]


def main():
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    ta, pe, _ = gg.load_reference()
    for (B, T, C, H, W, dffn, stride) in SHAPES:
        layer = ta.TemporalAxialTrajectoryAttentionLayer(d_model=C, d_ffn=dffn, n_heads=8).eval()
        seed = 15000 + T * 100 + H
        shapes, w = gg.load_random(layer, seed)
        g = torch.Generator().manual_seed(seed + 1)
        x = torch.randn(B, T, C, H, W, generator=g)
        src = x.permute(0, 1, 3, 4, 2).reshape(B * T, H * W, C).contiguous()
        pos = pe.PositionEmbeddingSine3D(C // 2, normalize=True)(x, fmt="btchw").permute(0, 1, 3, 4, 2).contiguous()
        out, ha, wa = layer(src, pos)
        gg.save(f"g15_shipped_B{B}_T{T}_C{C}_H{H}_W{W}",
                meta=gg.meta(B=B, T=T, C=C, H=H, W=W, d_ffn=dffn, heads=8, seed=seed, shapes=shapes, stride=stride),
                wsum=np.float64(gg.wsum(w)), out=out[:, ::stride], out_checks=gg.checks(out),
                h_attn_checks=gg.checks(ha), w_attn_checks=gg.checks(wa), pos_checks=gg.checks(pos))
        print(f"g15_shipped_B{B}_T{T}_C{C}_H{H}_W{W}: out {tuple(out.shape)} stored every {stride}th row")


if __name__ == "__main__":
    main()
