"""Batch sharding of the within-clip path over the GPUs of one node.

No op of TemporalAxialTrajectoryAttentionLayer mixes different clips (attention runs inside (b, w) / (b, h)
sequences, LayerNorm / FFN are per token), so the path shards over B with NO data-path collective: every rank runs the
layer on its own clips.  The only communication is the optional reassembly of the output map with one RCCL all-gather
(``torch.distributed`` backend "nccl" on ROCm), contiguous because dim 0 of [(B T), (H W), C] is the shard dimension.

Process model: one process per GPU (torchrun); this module never creates process groups itself.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist
from torch import Tensor


def shard_bounds(n_clips: int, world: int) -> List[Tuple[int, int]]:
    """Balanced contiguous split of `n_clips` over `world` ranks: [(start, end)] per rank (earlier ranks get the extra)."""
    q, r = divmod(n_clips, world)
    out, s = [], 0
    for k in range(world):
        e = s + q + (1 if k < r else 0)
        out.append((s, e))
        s = e
    return out


def local_slice(src: Tensor, pos: Tensor, rank: int, world: int) -> Tuple[Tensor, Tensor]:
    """Cut this rank's clips out of replicated inputs.  src [(B T), (H W), C], pos [B, T, H, W, C]."""
    B, T = pos.shape[:2]
    s, e = shard_bounds(B, world)[rank]
    p_loc = pos[s:e].contiguous()
    tag = getattr(pos, "_axvs_sine3d", None)
    if tag is not None and tag.version == pos._version and e > s:
        # the sine embedding does not depend on the clip index: the slice keeps the specification (axial_vs_amd.modules.SineTag)
        from .modules import tag_sine3d
        tag_sine3d(p_loc, tag.temperature, tag.normalize, tag.scale, tag.level)
    return src[s * T:e * T].contiguous(), p_loc


def gather_clips(out_local: Tensor, n_clips_total: int, frames: int, group=None, async_op: bool = False):
    """All-gather the per-rank outputs [(B_r T), (H W), C] into [(B T), (H W), C] on every rank.

    Equal shards use one `all_gather_into_tensor` (a single contiguous RCCL all-gather); ragged shards fall back to
    the list form.  With ``async_op`` the (work, tensor) pair is returned so the caller can overlap it with compute.
    """
    world = dist.get_world_size(group)
    bounds = shard_bounds(n_clips_total, world)
    sizes = [(e - s) * frames for s, e in bounds]
    tail = tuple(out_local.shape[1:])
    if len(set(sizes)) == 1:
        full = out_local.new_empty((sum(sizes),) + tail)
        work = dist.all_gather_into_tensor(full, out_local.contiguous(), group=group, async_op=async_op)
        return (work, full) if async_op else full
    # ragged split (B not a multiple of the world size): collectives need equal counts -> pad to the largest shard
    if async_op:
        raise NotImplementedError("async gather needs equal shards (B a multiple of the world size)")
    mx = max(sizes)
    padded = out_local.new_zeros((mx,) + tail)
    padded[: out_local.shape[0]] = out_local
    buf = out_local.new_empty((world * mx,) + tail)
    dist.all_gather_into_tensor(buf, padded, group=group)
    return torch.cat([buf[k * mx: k * mx + sizes[k]] for k in range(world)], dim=0)


def sharded_forward(layer_fn: Callable[[Tensor, Tensor], Tensor], src: Tensor, pos: Tensor, group=None,
                    gather: bool = True, replicated_inputs: bool = True) -> Tensor:
    """Run `layer_fn(src_local, pos_local) -> out_local` on this rank's clips and (optionally) reassemble the output.

    ``replicated_inputs``: src / pos hold the whole batch on every rank (cut locally); otherwise they are already the
    local shard and ``pos.shape[0]`` is the local clip count (the total is summed over ranks).
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    T = pos.shape[1]
    if replicated_inputs:
        total = pos.shape[0]
        s_loc, p_loc = local_slice(src, pos, rank, world)
    else:
        s_loc, p_loc = src, pos
        cnt = torch.tensor([pos.shape[0]], device=src.device, dtype=torch.int64)
        if world > 1:
            dist.all_reduce(cnt, group=group)
        total = int(cnt.item())
    out_local = layer_fn(s_loc, p_loc) if s_loc.shape[0] else s_loc.new_empty(s_loc.shape)
    if not gather or world == 1:
        return out_local
    return gather_clips(out_local, total, T, group)
