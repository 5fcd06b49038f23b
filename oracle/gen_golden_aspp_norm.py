"""tests/golden/g16_*: the cross-clip tracking module with ASPP norm_fn = 'syncbn' (the one alternative the reference's ConvBN can be built with besides the
shipped 'ln': kmax_pixel_decoder.py:32-40), eval mode, from the REFERENCE module (imported from /root/reference; build container only):

    python oracle/gen_golden_aspp_norm.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402  (the reference loader and the fixture helpers)


def main():
    torch.manual_seed(0)
    torch.set_grad_enabled(False)
    _, _, cc = gg.load_reference()
    for (Bv, Q, Tc) in [(2, 16, 3), (1, 16, 4)]:
        a = cc.ASPP(256, 256, [3, 3, 3], [1, 2, 3], 0.0, "syncbn").eval()
        shapes, w = gg.load_random(a, 7200 + Tc)
        g = torch.Generator().manual_seed(7300 + Tc)
        x = torch.randn(Bv * Q, 256, Tc, generator=g)
        gg.save(f"g16_cc_aspp_syncbn_BQ{Bv * Q}_Tc{Tc}", meta=gg.meta(BQ=Bv * Q, Tc=Tc, seed=7200 + Tc, shapes=shapes), wsum=np.float64(gg.wsum(w)), x=x, out=a(x))
    for (Bv, Q, Tc, V, Hh, Ww, layers, K) in [(1, 16, 3, 2, 8, 8, 2, 19), (1, 24, 4, 2, 6, 10, 3, 11)]:
        m = cc.CrossClipTrackingModule(num_layers=layers, num_classes=K, attn_drop=0.0, aspp_drop=0.0, kernel_sizes=[3, 3, 3], atrous_rates=[1, 2, 3],
                                       norm_fn="syncbn", num_clip_frames=V).eval()
        seed = 7000 + Tc * 10 + layers
        shapes, w = gg.load_random(m, seed)
        g = torch.Generator().manual_seed(seed + 1)
        cq = torch.randn(Bv, Q, Tc, 256, generator=g)
        pf = torch.nn.functional.normalize(torch.randn(Bv, 128, Tc * V, Hh, Ww, generator=g), dim=1)
        out = m(cq, pf)
        gg.save(f"g16_cc_module_syncbn_Q{Q}_Tc{Tc}_V{V}_H{Hh}_L{layers}",
                meta=gg.meta(B=Bv, Q=Q, Tc=Tc, V=V, H=Hh, W=Ww, layers=layers, num_classes=K, seed=seed, shapes=shapes, norm_fn="syncbn"),
                wsum=np.float64(gg.wsum(w)), pred_logits=out["pred_logits"], masks_checks=gg.checks(out["pred_masks"]), pred_masks=out["pred_masks"],
                aux0_logits=out["aux_outputs"][0]["pred_logits"], aux0_masks=out["aux_outputs"][0]["pred_masks"])


if __name__ == "__main__":
    main()
