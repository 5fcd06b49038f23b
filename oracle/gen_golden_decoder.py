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
    only_full = "--full-only" in sys.argv
    for (B, T, chans, sizes, stages, tl, seed, dffn, store_all, *rest) in [
            (1, 2, {"res3": 64, "res4": 96, "res5": 128}, {"res3": (16, 16), "res4": (8, 8), "res5": (4, 4)}, 2, 2, 81, 512, True),   # toy sizes: every output stored whole
            (1, 3, {"res3": 32, "res4": 64, "res5": 64}, {"res3": (12, 20), "res4": (6, 10), "res5": (3, 5)}, 1, 2, 82, 512, True),
            # temporal-only decoder (SPATIAL_LAYERS 0 -> TemporalTransformerEncoder, WC/msdeformattn.py:59-61,276-290)
            (1, 2, {"res3": 32, "res4": 64, "res5": 96}, {"res3": (16, 24), "res4": (8, 12), "res5": (4, 6)}, 2, 1, 84, 256, True, "temporal_only"),
            # BASELINE config 3 at full size: ConvNeXt-T pyramid of a 512 x 512 clip of T = 4 frames, the shipped
            # stage layout (NUM_STAGES 2, SPATIAL_LAYERS 2, TEMPORAL_LAYERS 4: configs/VIPSeg/.../maxtron_wc_*.yaml), d_ffn 1024.
            # Outputs are stored as strided subsamples + float64 checksums; so are the outputs of every stage's spatial layer and
            # temporal encoders (forward hooks on the reference modules), to localise a deviation.
            (1, 4, {"res3": 192, "res4": 384, "res5": 768}, {"res3": (64, 64), "res4": (32, 32), "res5": (16, 16)}, 2, 2, 83, 1024, False)]:
        if only_full and store_all:
            continue
        temporal_only = "temporal_only" in rest
        if "--temporal-only" in sys.argv and not temporal_only:
            continue
        strides = {"res3": 8, "res4": 16, "res5": 32}
        shape = {k: ShapeSpec(channels=c, stride=strides[k]) for k, c in chans.items()}
        m = ms.MSDeformAttnPixelDecoder(shape, transformer_dropout=0.0, transformer_attn_drop=0.0, transformer_nheads=8,
                                        transformer_dim_feedforward=dffn, transformer_num_stages=stages,
                                        transformer_spatial_layers=0 if temporal_only else stages, transformer_temporal_layers=stages * tl,
                                        transformer_temporal_attn_type="axial-trajectory", conv_dims=256,
                                        transformer_spatial_in_features=["res3", "res4", "res5"],
                                        transformer_temporal_in_features=["res4", "res5"], num_clip_frames=T,
                                        cross_clip_training=not store_all).eval()
        shp, w = load_random(m, seed)
        g = torch.Generator().manual_seed(seed + 1)
        feats = {k: torch.randn(B * T, chans[k], *sizes[k], generator=g) for k in chans}
        trace = []
        if not store_all:
            enc = m.transformer.encoder
            for i in range(stages):
                enc.spatial_layers[i].register_forward_hook(lambda mod, a, o, i=i: trace.append((f"s{i}_spatial", o)))
                enc.temporal_layers[i].register_forward_hook(lambda mod, a, o, i=i: trace.append((f"s{i}_temporal", o[0])))
        with torch.no_grad():
            out, _, _ = m.forward_features({k: v.clone() for k, v in feats.items()})
        mt = dict(B=B, T=T, chans=chans, sizes=sizes, stages=stages, temporal_per_stage=tl, d_ffn=dffn, seed=seed, shapes=shp)
        arrs = dict(wsum=np.float64(wsum(w)))
        if temporal_only:
            mt["temporal_only"] = True
        if store_all:
            for k, v in out.items():
                arrs["out_" + k] = v
            name = f"g8_pixel_decoder_T{T}_S{stages}" + ("_temporal_only" if temporal_only else "")
        else:
            sub = {"res3": 8, "res4": 4, "res5": 2}            # spatial stride of the stored subsample (every 4th channel)
            mt.update(sub=sub, csub=4, full_size=True)
            for k, v in out.items():
                arrs["out_" + k] = v[:, ::4, ::sub[k], ::sub[k]].contiguous()
                arrs["chk_" + k] = checks(v)
            seen = {}
            for key, ten in trace:                              # temporal hook fires per level: res5 first, then res4
                n = seen.get(key, 0)
                seen[key] = n + 1
                tag = key if "spatial" in key else key + ("_res5" if n == 0 else "_res4")
                arrs["tr_" + tag] = ten[:, ::37, ::4].contiguous()
                arrs["trchk_" + tag] = checks(ten)
            name = f"g8_pixel_decoder_full_T{T}_S{stages}"
        arrs["meta"] = meta(**mt)
        save(name, **arrs)


if __name__ == "__main__":
    main()
