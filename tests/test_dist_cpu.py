"""CPU, world_size 2, gloo: the batch-sharding + all-gather logic of axial_vs_amd.dist.
The HIP layer cannot run here, so the per-rank compute is the oracle (test infrastructure); what is under test is the
slicing, the ragged split and the reassembly order."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import axvs_oracle as orc


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


CASES = [(4, True), (3, True), (3, False), (1, True)]   # (clips, inputs replicated on every rank?)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from axial_vs_amd import dist as axd
        T, C, H, W, F = 2, 64, 4, 5, 128
        w = orc.random_weights(orc.axial_layer_param_shapes(C, F), 3)
        fn = lambda s, p: orc.axial_layer(s, p, w, 8, want_attn=False)[0]
        for B, replicated in CASES:
            src, pos = orc.synthetic_clip(B, T, C, H, W, seed=5)
            if replicated:
                out = axd.sharded_forward(fn, src, pos, gather=True, replicated_inputs=True)
            else:
                s_loc, p_loc = axd.local_slice(src, pos, rank, world)
                out = axd.sharded_forward(fn, s_loc, p_loc, gather=True, replicated_inputs=False)
            ref = fn(src, pos)
            q.put((rank, B, float((out - ref).abs().max()), tuple(out.shape)))
        # chunked, overlapped gather (groups of clips land in their places of the full map) and the 16-bit map
        src, pos = orc.synthetic_clip(8, T, C, H, W, seed=6)
        ref = fn(src, pos)
        for chunks, dt in ((2, None), (4, None), (2, torch.bfloat16), (1, torch.float16)):
            out = axd.sharded_forward(fn, src, pos, gather=True, chunks=chunks, gather_dtype=dt)
            want = ref if dt is None else ref.to(dt)
            assert out.dtype == want.dtype
            q.put((rank, 8, float((out.float() - want.float()).abs().max()), tuple(out.shape)))
        # pre-sharded inputs, chunked: every group's gather lands in ONE contiguous slice.  Default: natural (rank-major) clip order on
        # every path; allow_permuted=True: ALWAYS (map, order) -- (group, rank, clip) order without the reordering copy here ...
        s_loc, p_loc = axd.local_slice(src, pos, rank, world)
        out = axd.sharded_forward(fn, s_loc, p_loc, gather=True, replicated_inputs=False, chunks=2)
        assert isinstance(out, torch.Tensor)
        q.put((rank, 8, float((out - ref).abs().max()), tuple(out.shape)))
        out, order = axd.sharded_forward(fn, s_loc, p_loc, gather=True, replicated_inputs=False, chunks=2, allow_permuted=True)
        assert order == axd.chunked_clip_order(8, world, 2) and sorted(order) == list(range(8)) and order != list(range(8))
        want = ref.reshape(8, T, H * W, C)[order].reshape(8 * T, H * W, C)
        q.put((rank, 8, float((out - want).abs().max()), tuple(out.shape)))
        # ... and the identity order where the chunked path does not apply: a local clip count that `chunks` does not divide (4 per rank, 3 groups)
        for kw in (dict(), dict(allow_permuted=True)):
            got = axd.sharded_forward(fn, s_loc, p_loc, gather=True, replicated_inputs=False, chunks=3, **kw)
            if kw:
                assert got[1] == list(range(8))
                got = got[0]
            q.put((rank, 8, float((got - ref).abs().max()), tuple(got.shape)))
        # a layer_fn that already returns the 16-bit map (the kernel epilogue's out_dtype) is not cast again
        fn16 = lambda s, p: fn(s, p).to(torch.float16)
        out = axd.sharded_forward(fn16, src, pos, gather=True, chunks=2, gather_dtype=torch.float16)
        q.put((rank, 8, float((out.float() - ref.to(torch.float16).float()).abs().max()), tuple(out.shape)))
    finally:
        dist.destroy_process_group()


def test_sharded_forward_matches_unsharded():
    """One pair of gloo ranks runs every case (process start-up dominates the cost)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world * (len(CASES) + 9))]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert len(res) == world * (len(CASES) + 9)
    for rank, B, err, shape in res:
        assert shape == (B * 2, 20, 64)
        assert err < 1e-5, (rank, B, err)   # sharding never mixes clips (the 16-bit maps are compared with the cast reference)


def test_shard_bounds():
    from axial_vs_amd.dist import shard_bounds
    assert shard_bounds(64, 8) == [(8 * k, 8 * k + 8) for k in range(8)]
    assert shard_bounds(3, 2) == [(0, 2), (2, 3)]
    assert shard_bounds(1, 4) == [(0, 1), (1, 1), (1, 1), (1, 1)]


def _offaxis_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from axial_vs_amd import dist as axd
        B, T, C, H, W, F = 1, 2, 64, 6, 4, 128
        w = orc.random_weights(orc.axial_layer_param_shapes(C, F), 8)
        src, pos = orc.synthetic_clip(B, T, C, H, W, seed=8)
        fn = lambda x, p, which: orc.axial_pass(x, p, w, which)
        out = axd.offaxis_forward(fn, src, pos, gather=True)
        rows = axd.offaxis_forward(fn, src, pos, gather=False)
        ref = orc.axial_layer(src, pos, w, 8, want_attn=False)[0]
        hb = H // world
        ref_rows = ref.reshape(B, T, H, W, C)[:, :, rank * hb:(rank + 1) * hb]
        q.put((rank, float((out - ref).abs().max()), float((rows - ref_rows).abs().max()), tuple(out.shape)))
    finally:
        dist.destroy_process_group()


def test_offaxis_forward_matches_unsharded():
    """SURVEY 8e option (ii): ONE clip over 2 ranks -- height pass on column blocks, all-to-all, width pass + FFN on row blocks.
    The per-rank compute is the oracle's two passes (test infrastructure); under test: the block cuts, the exchange and the
    reassembly.  (gloo has no all-to-all: `_exchange` falls back to an all-gather.)"""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_offaxis_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, err, err_rows, shape in res:
        assert shape == (2, 24, 64)
        assert err < 1e-5 and err_rows < 1e-5, (rank, err, err_rows)


def test_oracle_axial_pass_composes_to_the_layer():
    B, T, C, H, W, F = 2, 3, 64, 5, 7, 128
    w = orc.random_weights(orc.axial_layer_param_shapes(C, F), 9)
    src, pos = orc.synthetic_clip(B, T, C, H, W, seed=9)
    ref = orc.axial_layer(src, pos, w, 8, want_attn=False)[0]
    y = orc.axial_pass(orc.axial_pass(src.reshape(B, T, H, W, C), pos, w, 0), pos, w, 1)
    assert float((y.reshape(B * T, H * W, C) - ref).abs().max()) < 1e-5


def test_no_list_form_all_gather_on_the_batch_sharded_path():
    """round-4 review: ProcessGroupNCCL flattens list-form all_gather into a temporary and copies every slice out -- the batch-sharded
    path must only use all_gather_into_tensor on contiguous destinations (the gloo stand-in of the all-to-all keeps its own)."""
    import inspect
    from axial_vs_amd import dist as axd
    for f in (axd.sharded_forward, axd.gather_clips, axd.offaxis_forward):
        assert "dist.all_gather(" not in inspect.getsource(f), f.__name__
