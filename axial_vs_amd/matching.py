"""Clip-to-clip query alignment on the device.

Reference: MaXTron_Video-kMaX/maxtron_deeplab/maxtron_cc_model.py:280-301 (the per-video loop that aligns every clip's
queries to the previous clip's) and :360-369 (`match_from_embds`: cosine cost + scipy.optimize.linear_sum_assignment on the
CPU); Tube-Link: models/video/tube_link_vis/mask2former_video_cc_head.py:907-913, :1038-1050.
Here the cost matrix and the assignment run in libaxvs.so and the indices stay on the GPU: no `.cpu()` sync per clip pair.
"""
from __future__ import annotations

import torch
from torch import Tensor

from . import _lib
from .modules import _dev_f32, _stream, _workspace, _guarded


@_guarded
def linear_sum_assignment(cost: Tensor) -> Tensor:
    """cost fp32 [n,n] or [batch,n,n] (CUDA) -> int64 column index per row = scipy.optimize.linear_sum_assignment(cost)[1]."""
    c = _dev_f32(cost, "cost")
    squeeze = c.dim() == 2
    if squeeze:
        c = c[None]
    b, n, m = c.shape
    if n != m:
        raise NotImplementedError("axial_vs_amd: only square assignment problems (query sets of equal size) are built")
    out = torch.empty(b, n, dtype=torch.int64, device=c.device)
    _lib.check(_lib.lib().axvs_linear_sum_assignment(c.data_ptr(), out.data_ptr(), b, n, _stream(c.device)), "axvs_linear_sum_assignment")
    return out[0] if squeeze else out


@_guarded
def match_from_embds(tgt_embds: Tensor, cur_embds: Tensor) -> Tensor:
    """maxtron_cc_model.py:360-369: permutation (int64, on the device) that makes `cur_embds` align with `tgt_embds`."""
    t = _dev_f32(tgt_embds, "tgt_embds")
    c = _dev_f32(cur_embds, "cur_embds")
    if t.shape != c.shape or t.dim() != 2:
        raise RuntimeError(f"tgt_embds {tuple(t.shape)} and cur_embds {tuple(c.shape)} must be equal [Q, C] matrices")
    Q, Cc = t.shape
    L = _lib.lib()
    ws = _workspace(t.device, L.axvs_match_embds_workspace_bytes(Q, Cc))
    idx = torch.empty(Q, dtype=torch.int64, device=t.device)
    _lib.check(L.axvs_match_embds(t.data_ptr(), c.data_ptr(), idx.data_ptr(), Q, Cc, ws.data_ptr(), ws.numel(), _stream(t.device)),
               "axvs_match_embds")
    return idx


@_guarded
def match_clips(pred_mask_embeddings: Tensor, pred_cluster_centers: Tensor) -> Tensor:
    """maxtron_cc_model.py:280-301: per video, align the queries of clip i to the already aligned clip i-1 by their mask
    embeddings and carry the cluster centres along.  pred_mask_embeddings / pred_cluster_centers [B, Tc, Q, C*] ->
    matched cluster centres [B, Q, Tc, C] (the `clip_query` input of CrossClipTrackingModule)."""
    emb = _dev_f32(pred_mask_embeddings, "pred_mask_embeddings")
    cen = pred_cluster_centers
    B, Tc, Q, Cc = emb.shape
    if Tc == 1:
        return cen.permute(0, 2, 1, 3).contiguous()
    L = _lib.lib()
    ws = _workspace(emb.device, L.axvs_match_clips_workspace_bytes(B, Tc, Q, Cc))
    idx = torch.empty(B, Tc - 1, Q, dtype=torch.int64, device=emb.device)
    _lib.check(L.axvs_match_clips(emb.data_ptr(), idx.data_ptr(), B, Tc, Q, Cc, ws.data_ptr(), ws.numel(), _stream(emb.device)),
               "axvs_match_clips")
    # carry the cluster centres along: clip 0 as is, clip i permuted by its alignment (plumbing: one gather)
    aligned = torch.cat([cen[:, :1], torch.gather(cen[:, 1:], 2, idx[..., None].expand(-1, -1, -1, cen.shape[-1]))], dim=1)
    return aligned.permute(0, 2, 1, 3).contiguous()
