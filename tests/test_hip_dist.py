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
        # direct all-gather by peer writes: the ranks exchange IPC handles of their output maps (hipIpcGetMemHandle through torch's CUDA IPC) and
        # write their rows straight into each other's maps -- no collective; bit-equal to the gathered forward, fp32 and 16-bit maps, chunked
        pm = axd.PeerMaps(B * T, (H * W, C), torch.float32, "cuda")
        res["peer_maps"] = bool(torch.equal(axd.sharded_forward(fn, s, p, peer_maps=pm), full))
        res["peer_maps_chunks2_presharded"] = bool(torch.equal(axd.sharded_forward(fn, s_loc, p_loc, replicated_inputs=False, chunks=2, peer_maps=pm), full))
        pm16 = axd.PeerMaps(B * T, (H * W, C), torch.float16, "cuda")
        res["peer_maps_f16"] = bool(torch.equal(axd.sharded_forward(fn, s, p, gather_dtype=torch.float16, chunks=2, peer_maps=pm16), full.half()))
        # two maps in turn: step 1's map still holds step 1's result after step 2 (other inputs) has been published by both ranks
        s2 = s * 0.5 + 0.25
        full2 = fn(s2, p)
        m1 = axd.sharded_forward(fn, s, p, peer_maps=pm16, gather_dtype=torch.float16)
        m2 = axd.sharded_forward(fn, s2, p, peer_maps=pm16, gather_dtype=torch.float16)
        res["peer_maps_two_buffers"] = bool(m1.data_ptr() != m2.data_ptr() and torch.equal(m1, full.half()) and torch.equal(m2, full2.half()))
        del pm, pm16
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


def _syncbn_worker(rank, world, port, q):
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
    from golden_util import rel_err
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        B, Q, Tc, V, H, W, nl, K = 1, 16, 3, 2, 8, 8, 2, 7
        w = orc.random_weights(orc.cc_module_param_shapes(nl, K), 91)
        cqs, pfs, dls, dms = [], [], [], []
        for r in range(world):                        # every rank knows every rank's (seeded) data: the expectation is the joint graph
            g = torch.Generator().manual_seed(900 + r)
            cqs.append(torch.randn(B, Q, Tc, 256, generator=g) + 0.3 * r)
            pfs.append(torch.nn.functional.normalize(torch.randn(B, 128, Tc * V, H, W, generator=g), dim=1))
            dls.append([torch.randn(1, Q, K + 1, generator=g) for _ in range(nl)])
            dms.append([torch.randn(B, Q, Tc * V, H, W, generator=g) * 0.05 for _ in range(nl)])
        wd = {k: v.double().requires_grad_("running" not in k) for k, v in w.items()}
        qd = [c.double().requires_grad_(True) for c in cqs]
        lg, mk, stats = orc.cc_module_train(qd, [p.double() for p in pfs], wd, nl, V, (1, 2, 3), 0.1, 0.1, 4242)
        loss = sum((a * b.double()).sum() for r in range(world) for a, b in zip(lg[r], dls[r])) + \
            sum((a * b.double()).sum() for r in range(world) for a, b in zip(mk[r], dms[r]))
        loss.backward()
        mod = ax.CrossClipTrackingModule(num_layers=nl, num_classes=K, attn_drop=0.1, aspp_drop=0.1, kernel_sizes=[3, 3, 3],
                                         atrous_rates=[1, 2, 3], norm_fn="ln", num_clip_frames=V)
        sd = mod.state_dict()
        sd.update(w)
        mod.load_state_dict(sd, strict=True)
        mod = mod.cuda().train()
        mod.dropout_seed = 4242
        cq = cqs[rank].cuda().requires_grad_(True)
        out = mod(cq, pfs[rank].cuda())
        logits = [a["pred_logits"] for a in out["aux_outputs"]] + [out["pred_logits"]]
        masks = [a["pred_masks"] for a in out["aux_outputs"]] + [out["pred_masks"]]
        (sum((a * b.cuda()).sum() for a, b in zip(logits, dls[rank])) + sum((a * b.cuda()).sum() for a, b in zip(masks, dms[rank]))).backward()
        res = {}
        res["logits"] = rel_err(torch.stack(logits).detach().cpu(), torch.stack(lg[rank]).detach())
        res["masks"] = rel_err(torch.stack(masks).detach().cpu(), torch.stack(mk[rank]).detach())
        res["d_clip_query"] = rel_err(cq.grad.cpu(), qd[rank].grad)
        # parameter gradients: each rank holds its own share (as under DDP before the averaging); their sum is the joint gradient
        scale = max(float(v.grad.norm()) for v in wd.values() if v.requires_grad)
        worst = 0.0
        for k, p in mod.named_parameters():
            gsum = p.grad.detach().clone()
            dist.all_reduce(gsum)
            ref = wd[k].grad
            worst = max(worst, float((gsum.cpu().double() - ref).norm() / max(float(ref.norm()), 1e-3 * scale)))
        res["param_grads"] = worst
        # running statistics: the joint batch statistics, identical on both ranks
        for name, per_layer in stats.items():
            rm, rv = w[name + ".running_mean"].double(), w[name + ".running_var"].double()
            for mean, var in per_layer:
                rm, rv = 0.99 * rm + 0.01 * mean, 0.99 * rv + 0.01 * var
            bufs = dict(mod.named_buffers())
            res["rm." + name] = rel_err(bufs[name + ".running_mean"].cpu(), rm)
            res["rv." + name] = rel_err(bufs[name + ".running_var"].cpu(), rv)
        torch.cuda.synchronize()
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def test_cross_clip_training_sync_batch_norm_over_two_ranks():
    """CrossClipTrackingModule.train() on two ranks with different clips: the library's BatchNorm partial sums travel through the
    all-reduce hook (gloo here, RCCL on a multi-GPU node), so outputs, input gradients, running statistics and the sum of the ranks'
    parameter gradients equal the float64 restatement whose BatchNorm runs over both ranks' rows (nn.SyncBatchNorm semantics)."""
    import __graft_entry__ as ge
    ge.build()
    assert torch.cuda.is_available()
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_syncbn_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, r in res:
        print(rank, {k: f"{v:.1e}" for k, v in r.items()})
        assert max(r.values()) < 1e-4, (rank, r)


def test_bench_launcher_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2` as the driver runs it at N > 1 -- the self-launch into torch.distributed.run, one rank per process,
    barrier + max-over-ranks timing, rank 0's single JSON line -- with both ranks on cuda:0 over gloo (AXVS_BENCH_SHARE_GPU=1: the
    box has one GPU; on a node the same command runs one rank per GPU over RCCL)."""
    import json
    import subprocess
    import sys
    import __graft_entry__ as ge
    ge.build()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, AXVS_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "10", "--warmup", "3",
           "--no-cpu-baseline", "--settle-ms", "20"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]            # rank 0 alone prints
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 10 and j["warmup"] == 3 and j["scaling"] == "weak" and j["value"] > 0
    assert abs(j["value"] - 2 * 4 / (j["ms_per_step"] * 1e-3)) / j["value"] < 1e-3      # whole-job frames/s: both ranks' clips / max-over-ranks time
    print("bench.py --gpus 2 (two ranks sharing cuda:0 over gloo):", j["value"], j["unit"], j["ms_per_step"], "ms per step")
    # what the first 8-GPU run will be read by: the world size torch.distributed reports, and BASELINE config 5 as worded -- the
    # [8,4,256,96,96] share per rank THROUGH the chunked all-gather of the output maps (fp32 and f16 maps) -- next to the
    # headline with `pos` as a plain tensor
    assert j["config"]["n_ranks_seen"] == 2 and "rccl_version" in j["config"]
    g5 = j["extras"]["cfg5_gather"]
    assert "error" not in g5, g5
    assert g5["ranks"] == 2 and g5["fp32_map"]["value"] > 0 and g5["f16_map"]["value"] > 0
    assert g5["fp32_map"]["inbound_MB_per_rank_per_step"] == pytest.approx(8 * 4 * 96 * 96 * 256 * 4 / 1e6, rel=1e-3)     # one other rank's maps
    assert g5["f16_map"]["inbound_MB_per_rank_per_step"] == pytest.approx(8 * 4 * 96 * 96 * 256 * 2 / 1e6, rel=1e-3)
    # the transport without a collective (dist.PeerMaps: every rank writes its rows into every rank's IPC-mapped map) ran as well
    pw = g5.get("f16_map_peer_writes")
    assert pw is not None and "error" not in pw and pw["value"] > 0, pw
    assert pw["inbound_MB_per_rank_per_step"] == pytest.approx(8 * 4 * 96 * 96 * 256 * 2 / 1e6, rel=1e-3)
    assert j["extras"]["tensor_pos"]["value"] > 0
    print("cfg5_gather:", g5["fp32_map"], g5["f16_map"], pw)


def test_bench_watchdog_prints_the_headline_when_a_multi_rank_extra_does_not_return():
    """N > 1: the secondary measurements of bench.py contain collectives; if one hangs, the headline must still reach the driver.  With a 2-second limit
    the watchdog fires during the extras: rank 0 prints the headline line (extras.watchdog says why it is alone), every rank exits 0."""
    import json
    import subprocess
    import sys
    import __graft_entry__ as ge
    ge.build()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, AXVS_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", AXVS_BENCH_EXTRAS_TIMEOUT_S="2")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--settle-ms", "20"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["value"] > 0 and "watchdog" in j["extras"], j.get("extras")
    print("watchdog line:", j["value"], j["unit"], "|", j["extras"]["watchdog"])
