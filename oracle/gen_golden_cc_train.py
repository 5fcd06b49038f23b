"""Generate tests/golden/g13_cc_train_*.npz: the REFERENCE CrossClipTrackingModule in train() mode under autograd.

CrossClipTrackingModule (CC/maxtron_cross_clip_tracking_module.py:204-331) is run in float64, train() mode -- the predictor's
training branch (:53-57), BatchNorm on batch statistics with its running-statistics update -- with every nn.Dropout replaced by a
module that multiplies by `axvs_oracle.dropout_keep(seed, site, ...)` (sites 10 + 2 l: attention maps of layer l, 11 + 2 l: the
ASPP's _proj_drop), so the fixture pins WHERE the reference applies dropout, not torch's RNG stream.  Saved: the outputs of every
layer (pred + aux), the gradients of clip_query and of every parameter for seeded upstream gradients on all outputs, and the
BatchNorm buffers after the forward.  Runs only in the build container.

    python oracle/gen_golden_cc_train.py
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402
from gen_golden_train import HashDropout  # noqa: E402

CONFIGS = [  # B, Q, Tc, V, H, W, layers, classes, p_attn, p_aspp
    (1, 16, 3, 2, 8, 8, 2, 19, 0.0, 0.0),
    (1, 16, 4, 2, 8, 8, 2, 19, 0.1, 0.2),
    (2, 8, 2, 1, 4, 8, 1, 5, 0.0, 0.0),
    (1, 24, 5, 1, 8, 8, 3, 7, 0.0, 0.1),
    (1, 8, 12, 1, 5, 7, 1, 5, 0.0, 0.1),      # 12 clips (NUM_VIDEO_FRAMES 24 / NUM_CLIP_FRAMES 2: the shipped training config), 35 pixels per clip (odd)
    (1, 16, 2, 2, 3, 5, 2, 7, 0.1, 0.0),      # 30 pixels per clip: rows 8-byte aligned only
]
ONLY = None   # set to a list of indices to regenerate a subset


def main():
    torch.manual_seed(0)
    _, _, cc = gg.load_reference()
    for ci, (B, Q, Tc, V, H, W, nl, K, p_attn, p_aspp) in enumerate(CONFIGS):
        if ONLY is not None and ci not in ONLY:
            continue
        m = cc.CrossClipTrackingModule(num_layers=nl, num_classes=K, attn_drop=p_attn, aspp_drop=p_aspp, kernel_sizes=[3, 3, 3],
                                       atrous_rates=[1, 2, 3], norm_fn="ln", num_clip_frames=V)
        seed = 13000 + Tc * 100 + Q + nl
        shapes, w = gg.load_random(m, seed)
        m = m.double().train()
        dseed = seed * 5 + 3
        for l in range(nl):
            m.transformer_trajectory_self_attention_layers[l].self_attn.attn_drop = HashDropout(p_attn, dseed, [10 + 2 * l])
            m.conv_short_aggregate_layers[l]._proj_drop = HashDropout(p_aspp, dseed, [11 + 2 * l])
        g = torch.Generator().manual_seed(seed + 1)
        cq = torch.randn(B, Q, Tc, 256, generator=g).double().requires_grad_(True)
        pf = torch.nn.functional.normalize(torch.randn(B, 128, Tc * V, H, W, generator=g), dim=1).double()
        out = m(cq, pf)
        logits = [a["pred_logits"] for a in out["aux_outputs"]] + [out["pred_logits"]]
        masks = [a["pred_masks"] for a in out["aux_outputs"]] + [out["pred_masks"]]
        d_logits = [torch.randn(t.shape, generator=g).double() for t in logits]
        d_masks = [torch.randn(t.shape, generator=g).double() * 0.05 for t in masks]
        loss = sum((a * b).sum() for a, b in zip(logits, d_logits)) + sum((a * b).sum() for a, b in zip(masks, d_masks))
        loss.backward()
        arrs = dict(meta=gg.meta(B=B, Q=Q, Tc=Tc, V=V, H=H, W=W, layers=nl, num_classes=K, seed=seed, dropout_seed=dseed, p_attn_drop=p_attn,
                                 p_aspp_drop=p_aspp, shapes=shapes),
                    wsum=np.float64(gg.wsum(w)), logits=torch.stack(logits).float(), masks=torch.stack(masks).float(),
                    d_logits=torch.stack(d_logits).float(), d_masks=torch.stack(d_masks).float(), d_clip_query=cq.grad.float())
        for k, v in m.named_parameters():      # big weight gradients: every 5th element + (sum, sum of squares, max |.|)
            gr = v.grad
            arrs["grad." + k] = gr.float() if gr.numel() <= 20000 else gr.reshape(-1)[::5].float()
            arrs["gradchk." + k] = gg.checks(gr)
        for k, v in m.named_buffers():
            arrs["buf." + k] = v.double().numpy() if v.dtype.is_floating_point else v.numpy()
        gg.save(f"g13_cc_train_B{B}_Q{Q}_Tc{Tc}_V{V}_H{H}_L{nl}", **arrs)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        ONLY = [int(a) for a in sys.argv[1:]]
    main()
