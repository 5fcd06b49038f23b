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


def gather_clips(out_local: Tensor, n_clips_total: int, frames: int, group=None, async_op: bool = False, dtype: Optional[torch.dtype] = None):
    """All-gather the per-rank outputs [(B_r T), (H W), C] into [(B T), (H W), C] on every rank.

    Equal shards use one `all_gather_into_tensor` (a single contiguous RCCL all-gather); ragged shards fall back to
    the list form.  With ``async_op`` the (work, tensor) pair is returned so the caller can overlap it with compute.
    ``dtype`` (torch.float16 / torch.bfloat16): the map is cast before it crosses the links and returned in that type -- half the
    bytes of the fp32 map (BASELINE config 5 is worded "bf16": 75 MB instead of 151 MB per rank).
    """
    world = dist.get_world_size(group)
    bounds = shard_bounds(n_clips_total, world)
    sizes = [(e - s) * frames for s, e in bounds]
    if dtype is not None and out_local.dtype != dtype:
        out_local = out_local.to(dtype)
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


class PeerMaps:
    """Direct all-gather by PEER WRITES (round 6): the transport that can meet the 8-GPU target where a ring cannot.

    xGMI is point-to-point: 7 links x ~153 GB/s per GPU.  A ring all-gather moves every rank's (world - 1) inbound shares over ONE inbound link
    (BASELINE config 5, f16 maps: 528 MB per rank and step = 3.4 ms against 1.95 ms of compute); written DIRECTLY by their producers the same bytes
    arrive over all seven links at once (0.49 ms).  Every rank owns one full output map; the ranks exchange the maps' IPC handles once
    (hipIpcGetMemHandle / hipIpcOpenMemHandle through torch's CUDA IPC -- the dmabuf mode of this image: HSA_ENABLE_IPC_MODE_LEGACY=0) and from then on
    `publish()` copies this rank's rows straight into EVERY rank's map: plain device-to-device copies (the SDMA engines / peer stores over xGMI), no
    collective launch, no staging buffer.  The copies run on a side stream behind the kernels that produced the rows, so a group's rows cross the links
    while the next group computes (`sharded_forward(..., peer_maps=...)`).  `wait()` = this rank's copies are done + a barrier: all ranks' rows have
    landed in this rank's map.  One process per GPU; works the same with both processes on ONE device (the test: two ranks sharing cuda:0).
    Not measured on more than one GPU (no multi-GPU node was available to this build): correctness by construction + the two-process test."""

    def __init__(self, rows_total: int, tail: Tuple[int, ...], dtype: torch.dtype, device, group=None, buffers: int = 2):
        """`buffers` maps used in turn, one per step (a step = the publishes up to a `wait()`).  Why two by default: a peer that has passed step i's
        barrier may publish step i + 1's rows while this rank's consumers of step i's map are still running -- into the SAME map that would be a race no
        stream order on this rank can prevent.  With two maps a peer writes into step i's map again only at step i + 2, i.e. after step i + 1's barrier,
        which this rank enters after its own step-(i + 1) copies have completed -- and those were queued behind everything this rank had queued on the
        publishing stream before, the consumers of step i's map included.  (Consumers on OTHER streams must be joined before the next step's first
        publish; `buffers=1` is for callers that synchronise themselves, e.g. a barrier of their own before each step.)"""
        from torch.multiprocessing.reductions import reduce_tensor
        if buffers < 1:
            raise ValueError("PeerMaps: buffers >= 1")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.device = torch.device(device)
        # blocks of their own from the caching allocator (IPC shares the whole allocation a tensor lives in)
        self.maps: List[Tensor] = [torch.empty((rows_total,) + tuple(tail), dtype=dtype, device=self.device) for _ in range(buffers)]
        handles = [None] * self.world
        dist.all_gather_object(handles, [reduce_tensor(m) for m in self.maps], group=group)
        self.peers: List[List[Tensor]] = []              # [buffer][rank]
        err = None
        try:
            for b in range(buffers):
                self.peers.append([self.maps[b] if r == self.rank else handles[r][b][0](*handles[r][b][1]) for r in range(self.world)])
        except Exception as e:                   # e.g. hipIpcOpenMemHandle refused (legacy IPC mode, no peer access): every rank has to learn of it,
            err = e                              # or the others would wait in the barrier below for a rank that has left
        oks = [None] * self.world
        dist.all_gather_object(oks, err is None, group=group)     # (also the barrier: every rank has opened every handle before anybody writes)
        if not all(oks):
            self.peers = []
            raise RuntimeError(f"PeerMaps: opening the peers' maps failed on rank(s) {[r for r, ok in enumerate(oks) if not ok]}"
                               + (f" (this rank: {err})" if err is not None else "")) from err
        self._cur = 0
        self._side = torch.cuda.Stream(self.device)

    @property
    def map(self) -> Tensor:
        """The map of the step being published (what the next `wait()` returns)."""
        return self.maps[self._cur]

    def publish(self, rows: Tensor, row0: int) -> None:
        """rows -> [row0, row0 + len(rows)) of every rank's map of this step, asynchronously behind the work already queued on the current stream."""
        cur = torch.cuda.current_stream(self.device)
        self._side.wait_stream(cur)
        with torch.cuda.stream(self._side):
            for k in range(self.world):
                r = (self.rank + k) % self.world         # start with my own map, then ring order: the ranks do not all hit the same peer first
                self.peers[self._cur][r][row0:row0 + rows.shape[0]].copy_(rows, non_blocking=True)
        rows.record_stream(self._side)

    def wait(self) -> Tensor:
        """All ranks' published rows of this step are in this rank's map (returns it; the next step publishes into the next map).  Host-synchronous:
        the step boundary of a batch-sharded forward."""
        self._side.synchronize()
        dist.barrier(self.group)
        torch.cuda.current_stream(self.device).wait_stream(self._side)
        full = self.maps[self._cur]
        self._cur = (self._cur + 1) % len(self.maps)
        return full


def chunked_clip_order(total: int, world: int, chunks: int) -> List[int]:
    """Clip index (rank-major numbering: rank r owns clips [r*b, (r+1)*b), b = total / world) at every position of the map that
    `sharded_forward(..., replicated_inputs=False, chunks=k)` returns: the map is ordered (group, rank, clip in group) so that every
    group's all-gather has ONE contiguous destination.  `full.view(chunks, world, cb * T, ...)[c, r]` is rank r's group c."""
    b = total // world
    cb = b // chunks
    return [r * b + c * cb + i for c in range(chunks) for r in range(world) for i in range(cb)]


def sharded_forward(layer_fn: Callable[[Tensor, Tensor], Tensor], src: Tensor, pos: Tensor, group=None,
                    gather: bool = True, replicated_inputs: bool = True, gather_dtype: Optional[torch.dtype] = None,
                    chunks: int = 1, allow_permuted: bool = False, peer_maps: Optional["PeerMaps"] = None):
    """Run `layer_fn(src_local, pos_local) -> out_local` on this rank's clips and (optionally) reassemble the output.

    ``replicated_inputs``: src / pos hold the whole batch on every rank (cut locally); otherwise they are already the
    local shard and ``pos.shape[0]`` is the local clip count (the total is summed over ranks).
    ``gather_dtype``: see `gather_clips`; a `layer_fn` that already returns that type (the layer's ``out_dtype``: the 16-bit map
    written by the kernel epilogue) is not cast again.
    ``chunks`` > 1 (equal shards only): the local clips are run in that many groups and every group's all-gather is issued
    asynchronously right behind its kernels, so the collective of group i crosses the links while group i + 1 computes.  Every group
    is ONE `all_gather_into_tensor` into a contiguous slice of the full map -- no list-form all_gather (ProcessGroupNCCL flattens
    those into a temporary and copies every slice out again: (world - 1) / world of the map in extra HBM reads and writes per step):
      * replicated inputs: the clips are dealt to the ranks group by group -- rank r computes clips {c * world * cb + r * cb + i}
        (cb = clips per group) -- so group c of all ranks IS rows [c * world * cb * T, (c + 1) * world * cb * T) of the map in natural
        clip order;
      * pre-sharded inputs (rank r holds its own clips): the groups land ordered (group, rank, clip in group).

    ``peer_maps`` (a `PeerMaps` of [total * T rows, ...] in the map's dtype; equal shards): no collective at all -- every group's rows are written
    straight into every rank's map at their natural (rank-major) position behind the group's kernels; the result is the step's map in natural clip
    order, valid until the same buffer comes round again (`PeerMaps(buffers=2)`: until the call after the next one).

    **The returned order never depends on the shapes.**  ``allow_permuted=False`` (default): a Tensor in natural clip order (rank-major for
    pre-sharded inputs) on every path -- the chunked pre-sharded path pays ONE reordering copy of the gathered map for it.
    ``allow_permuted=True``: ALWAYS a tuple ``(full, order)`` with ``order[i]`` = the clip (rank-major numbering) at position i of ``full``:
    `chunked_clip_order(total, world, chunks)` on the chunked pre-sharded path (no copy), the identity on every other path (replicated inputs,
    one rank, unequal shards, a local clip count that `chunks` does not divide).
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    T = pos.shape[1]
    tag = getattr(pos, "_axvs_sine3d", None)
    if tag is not None and tag.version != pos._version:
        tag = None

    def retag(p_c: Tensor) -> Tensor:
        if tag is None:
            return p_c
        from .modules import tag_sine3d
        return tag_sine3d(p_c.contiguous(), tag.temperature, tag.normalize, tag.scale, tag.level)

    equal = True
    if replicated_inputs:
        total = pos.shape[0]
    else:
        # every rank learns every rank's clip count (one tiny all-gather): the total, and whether the shards are equal -- all ranks
        # must take the same branch below
        mine = torch.tensor([pos.shape[0]], device=src.device, dtype=torch.int64)
        counts = mine.new_empty(world)
        if world > 1:
            dist.all_gather_into_tensor(counts, mine, group=group)
        else:
            counts = mine
        cl = counts.tolist()
        total = int(sum(cl))
        equal = len(set(cl)) == 1
    b_each = total // world if world else total
    if gather and world > 1 and peer_maps is not None:
        if not (equal and total % world == 0 and b_each > 0):
            raise RuntimeError("sharded_forward(peer_maps=...): equal shards only")
        ch = chunks if (chunks > 0 and b_each % chunks == 0) else 1
        cb = b_each // ch
        for c in range(ch):
            if replicated_inputs:                             # my own clips [rank * b_each, (rank + 1) * b_each), group by group
                c0 = rank * b_each + c * cb
                s_c, p_c = src[c0 * T:(c0 + cb) * T], retag(pos[c0:c0 + cb])
            else:
                s_c, p_c = src[c * cb * T:(c + 1) * cb * T], retag(pos[c * cb:(c + 1) * cb])
            o = layer_fn(s_c.contiguous(), p_c)
            if gather_dtype is not None and o.dtype != gather_dtype:
                o = o.to(gather_dtype)
            peer_maps.publish(o.contiguous(), (rank * b_each + c * cb) * T)
        full = peer_maps.wait()
        return (full, list(range(total))) if allow_permuted else full
    if gather and world > 1 and chunks > 1 and equal and total % world == 0 and b_each % chunks == 0 and b_each > 0:
        cb = b_each // chunks                                 # clips per group
        full = None
        works = []
        for c in range(chunks):
            if replicated_inputs:                             # group c of rank r: clips [c * world * cb + r * cb, ... + cb) of the batch
                c0 = c * world * cb + rank * cb
                s_c, p_c = src[c0 * T:(c0 + cb) * T], retag(pos[c0:c0 + cb])
            else:                                             # my own clips, in order
                s_c, p_c = src[c * cb * T:(c + 1) * cb * T], retag(pos[c * cb:(c + 1) * cb])
            o = layer_fn(s_c.contiguous(), p_c)
            if gather_dtype is not None and o.dtype != gather_dtype:
                o = o.to(gather_dtype)
            if full is None:
                full = o.new_empty((total * T,) + tuple(o.shape[1:]))
            dst = full[c * world * cb * T:(c + 1) * world * cb * T]        # contiguous: ranks side by side inside the group's slice
            works.append(dist.all_gather_into_tensor(dst, o.contiguous(), group=group, async_op=True))
        for wk in works:
            wk.wait()
        if replicated_inputs:
            return (full, list(range(total))) if allow_permuted else full
        if allow_permuted:
            return full, chunked_clip_order(total, world, chunks)
        # (group, rank, clip) -> (rank, group, clip): one copy of the map, the price of the invariant order
        tail = tuple(full.shape[1:])
        return full.view((chunks, world, cb * T) + tail).transpose(0, 1).reshape((total * T,) + tail)
    if replicated_inputs:
        s_loc, p_loc = local_slice(src, pos, rank, world)
    else:
        s_loc, p_loc = src, pos
    out_local = layer_fn(s_loc, p_loc) if s_loc.shape[0] else s_loc.new_empty(s_loc.shape)
    if not gather or world == 1:
        res = out_local if gather_dtype is None or out_local.dtype == gather_dtype else out_local.to(gather_dtype)
        n_here = total if (gather and world == 1) else out_local.shape[0] // max(T, 1)
        return (res, list(range(n_here))) if allow_permuted else res
    res = gather_clips(out_local, total, T, group, dtype=gather_dtype)
    return (res, list(range(total))) if allow_permuted else res


def _exchange(blocks: List[Tensor], group=None) -> List[Tensor]:
    """all-to-all of equally shaped blocks: blocks[j] goes to rank j, the result's entry i came from rank i.  RCCL / MPI:
    `all_to_all_single`; gloo (the CPU tests) has no all-to-all: an all-gather of everything, keeping my column (world times the
    bytes).  The path is chosen from the backend up front -- a failing collective propagates instead of being retried on a
    communicator that other ranks may still be using."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    send = torch.stack(blocks, 0).contiguous()
    if str(dist.get_backend(group)).lower() != "gloo":
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv, send, group=group)
        return list(recv.unbind(0))
    every = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(every, send, group=group)
    return [every[i][rank] for i in range(world)]


def offaxis_forward(pass_fn: Callable[[Tensor, Tensor, int], Tensor], src: Tensor, pos: Tensor, group=None, gather: bool = True) -> Tensor:
    """One clip (or a batch smaller than the GPU count) over several GPUs: SURVEY.md 8e option (ii).

    The height pass mixes tokens along H only -> rank r runs it on its block of COLUMNS; the width pass, norm1, FFN and norm2 mix
    along W / per token only -> rank r runs them on its block of ROWS; in between the [H, W_r] blocks are re-cut into [H_r, W]
    blocks with ONE all-to-all (the whole activation crosses the links once: B*T*H*W*C*4 bytes in total).  At the metric shape
    the exchange is 16.8 MB against ~50 us of compute per rank: latency-dominated -- this path exists for clips too large for
    one GPU or batches smaller than the node, not to speed up B = 1.

    `pass_fn(x [B,T,h,w,C], pos block, which)`: `layer.forward_pass`.  src [(B T),(H W),C] and pos [B,T,H,W,C] are replicated on
    every rank; H and W must be multiples of the world size.  Returns the full output [(B T),(H W),C] (gather=True) or this
    rank's rows [B,T,H_r,W,C]."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    B, T, H, W, C = pos.shape
    x = src.reshape(B, T, H, W, C)
    if world == 1:
        y = pass_fn(pass_fn(x, pos, 0), pos, 1)
        return y.reshape(B * T, H * W, C) if gather else y
    if H % world or W % world:
        raise RuntimeError(f"offaxis_forward: H={H} and W={W} must be multiples of the world size {world}")
    hb, wb = H // world, W // world
    w0, h0 = rank * wb, rank * hb
    y_cols = pass_fn(x[:, :, :, w0:w0 + wb].contiguous(), pos[:, :, :, w0:w0 + wb].contiguous(), 0)            # [B,T,H,wb,C]
    recv = _exchange([y_cols[:, :, j * hb:(j + 1) * hb].contiguous() for j in range(world)], group)              # from rank i: [B,T,hb,wb_i,C]
    y_rows = torch.cat(recv, dim=3)                                                                             # [B,T,hb,W,C]
    out_rows = pass_fn(y_rows, pos[:, :, h0:h0 + hb].contiguous(), 1)
    if not gather:
        return out_rows
    # one contiguous all-gather ([world, B, T, hb, W, C]); the row blocks are then interleaved into the map with ONE copy
    every = out_rows.new_empty((world * out_rows.shape[0],) + tuple(out_rows.shape[1:]))      # (concatenated form: gloo knows no other)
    dist.all_gather_into_tensor(every, out_rows.contiguous(), group=group)
    return every.view((world,) + tuple(out_rows.shape)).permute(1, 2, 0, 3, 4, 5).reshape(B * T, H * W, C)
