"""GPU, two processes sharing cuda:0 over a gloo process group: the REAL HIP layer through axial_vs_amd.dist (batch sharding with
the reassembling all-gather, chunked / 16-bit gathers, and one clip sharded off-axis with the all-to-all between the passes).
The box has one GPU, so both ranks launch on device 0; what differs from an 8-GPU RCCL run is the transport only (bench.py
--gpus N is the RCCL path).  Bit-equality with the unsharded call on the same rank is the bar: sharding never mixes clips."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "oracle"), os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    import axvs_oracle as orc
    import axial_vs_amd as ax
    from axial_vs_amd import dist as axd
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        B, T, C, H, W, F = 4, 2, 256, 16, 24, 512
        w = orc.random_weights(orc.axial_layer_param_shapes(C, F), 21)
        src, pos = orc.synthetic_clip(B, T, C, H, W, seed=21)
        layer = ax.TemporalAxialTrajectoryAttentionLayer(C, F, n_heads=8).eval()
        layer.load_state_dict(w, strict=True)
        layer = layer.cuda()
        s, p = src.cuda(), pos.cuda()
        fn = lambda a, b: layer(a, b)[0]
        full = fn(s, p)
        res = {}
        res["sharded"] = bool(torch.equal(axd.sharded_forward(fn, s, p), full))
        res["chunks2"] = bool(torch.equal(axd.sharded_forward(fn, s, p, chunks=2), full))
        res["f16_map"] = bool(torch.equal(axd.sharded_forward(fn, s, p, gather_dtype=torch.float16), full.half()))
        s_loc, p_loc = axd.local_slice(s, p, rank, world)
        res["presharded"] = bool(torch.equal(axd.sharded_forward(fn, s_loc, p_loc, replicated_inputs=False), full))
        # generated positions keep their specification through the slicing (the layer evaluates them in its loaders)
        pg = ax.PositionEmbeddingSine3D(C // 2, normalize=True).channels_last(B, T, H, W, "cuda")
        res["sine_tag"] = bool(torch.equal(axd.sharded_forward(fn, s, pg, chunks=2), fn(s, pg)))
        # one clip over both ranks: column blocks -> exchange -> row blocks
        one_s, one_p = s[:T].contiguous(), p[:1].contiguous()
        off = axd.offaxis_forward(layer.forward_pass, one_s, one_p, gather=True)
        res["offaxis"] = bool(torch.equal(off, fn(one_s, one_p)))
        torch.cuda.synchronize()
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def test_two_ranks_share_the_gpu_real_layer():
    import __graft_entry__ as ge
    ge.build()
    assert torch.cuda.is_available()
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, r in res:
        print(rank, r)
        assert all(r.values()), (rank, r)
