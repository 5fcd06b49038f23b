"""Golden vectors for the within-clip pixel decoder: the reference's own `MSDeformAttnPixelDecoder`
(WC/msdeformattn.py:293-437) imported from /root/reference; its CUDA extension is absent here, so MSDeformAttn takes the
reference's own PyTorch branch (see gen_golden_msda.py).

    python oracle/gen_golden_decoder.py      # writes tests/golden/g8_pixel_decoder_*.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import axvs_oracle as orc  # noqa: E402
import gen_golden  # noqa: E402
from gen_golden import _mod, load_random, wsum, meta, checks, save  # noqa: E402


def main():
    gen_golden.load_reference()
    WC = "maxtron_deeplab.modeling.within_clip_tracking_module"
    _mod("MultiScaleDeformableAttention")
    ms = gen_golden._load(WC + ".msdeformattn")
    ShapeSpec = sys.modules["detectron2.layers"].ShapeSpec
    for (B, T, chans, sizes, stages, tl, seed) in [(1, 2, {"res3": 64, "res4": 96, "res5": 128}, {"res3": (16, 16), "res4": (8, 8), "res5": (4, 4)}, 2, 2, 81),
                                                   (1, 3, {"res3": 32, "res4": 64, "res5": 64}, {"res3": (12, 20), "res4": (6, 10), "res5": (3, 5)}, 1, 2, 82)]:
        strides = {"res3": 8, "res4": 16, "res5": 32}
        shape = {k: ShapeSpec(channels=c, stride=strides[k]) for k, c in chans.items()}
        m = ms.MSDeformAttnPixelDecoder(shape, transformer_dropout=0.0, transformer_attn_drop=0.0, transformer_nheads=8,
                                        transformer_dim_feedforward=512, transformer_num_stages=stages,
                                        transformer_spatial_layers=stages, transformer_temporal_layers=stages * tl,
                                        transformer_temporal_attn_type="axial-trajectory", conv_dims=256,
                                        transformer_spatial_in_features=["res3", "res4", "res5"],
                                        transformer_temporal_in_features=["res4", "res5"], num_clip_frames=T,
                                        cross_clip_training=False).eval()
        shp, w = load_random(m, seed)
        g = torch.Generator().manual_seed(seed + 1)
        feats = {k: torch.randn(B * T, chans[k], *sizes[k], generator=g) for k in chans}
        with torch.no_grad():
            out, _, _ = m.forward_features({k: v.clone() for k, v in feats.items()})
        arrs = dict(meta=meta(B=B, T=T, chans=chans, sizes=sizes, stages=stages, temporal_per_stage=tl, d_ffn=512, seed=seed, shapes=shp),
                    wsum=np.float64(wsum(w)))
        for k, v in out.items():
            arrs["out_" + k] = v
        save(f"g8_pixel_decoder_T{T}_S{stages}", **arrs)


if __name__ == "__main__":
    main()
