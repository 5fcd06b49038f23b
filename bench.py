#!/usr/bin/env python3
"""Headline benchmark: axial-trajectory-attention forward, frames/s at [B=1,T=4,C=256,H=W=64] per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one TemporalAxialTrajectoryAttentionLayer forward (height pass + width pass + LN + FFN + LN,
WC/temporal_attention.py:187-220) over one clip resident in HBM.  With N ranks every rank owns its own clip
(the path shards over B with no data-path collective: SURVEY.md 8e) -> "scaling": "weak"; `--gather` adds the
north star's RCCL all-gather of the output maps, overlapped on a side stream.

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline     -- algorithmic FLOPs of the layer / mean forward time from HIP events on the launch stream,
                  against the dense 16-bit MFMA peak (2.5 PFLOP/s), and the per-kernel split of that time
  cpu_baseline -- the CPU oracle (torch port of the reference, oracle/axvs_oracle.py) timed on this host
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0  # dense bf16/f16, MI355X_MICROARCH.md "Chip-level parameters"


def layer_flops(B, T, H, W, C, F):
    """algorithmic FLOPs (2*MACs) of one layer in the reference's formulation."""
    def f_pass(S, L):
        N = T * L
        return S * (N * C * C * (10 + 4 * T) + 4 * N * N * C + 4 * N * T * C)
    return f_pass(B * W, H) + f_pass(B * H, W) + 4 * B * T * H * W * C * F


def layer_bytes(B, T, H, W, C, F, e=4):
    """src + pos read, out written (fp32 at the boundary) + 16-bit weights once."""
    P = 2 * (7 * C * C + 8 * C) + 2 * C * F + F + 5 * C
    return 3 * B * T * H * W * C * e + P * 2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16"])
    ap.add_argument("--shape", default="1,4,256,64,64", help="B,T,C,H,W per rank")
    ap.add_argument("--d-ffn", type=int, default=1024)
    ap.add_argument("--gather", action="store_true", help="all-gather every step's output across ranks (RCCL, side stream)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay a captured HIP graph per step instead of launching from Python")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--tensor-pos", action="store_true",
                    help="read `pos` as a plain tensor (default: pos comes from PositionEmbeddingSine3D and is evaluated in-kernel)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend; gloo + AXVS_BENCH_SHARE_GPU=1 runs N ranks on ONE GPU (launcher smoke test only)")
    ap.add_argument("--opt", action="append", default=[], help="library tuning option key=value (axvs_set_option)")
    args = ap.parse_args()

    import __graft_entry__ as ge
    if "RANK" not in os.environ and args.gpus > 1:
        # Invoked as `python bench.py --gpus N`: become the launcher.  This process has made no GPU call (torch is not even
        # imported yet); it compiles the library if needed, starts one worker per GPU under torch.distributed.run as a CHILD
        # process and relays its output and exit code (never exec: see the gpurun notes on re-exec after GPU initialisation).
        import socket
        import subprocess
        ge.build(load=False)
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    share = os.environ.get("AXVS_BENCH_SHARE_GPU") == "1" and args.backend == "gloo"
    dev = torch.device("cuda", 0 if share else local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    # one rank builds (a no-op when the in-tree .so is fresh), the others wait for it
    if local_rank == 0:
        ge.build()
    if world > 1:
        dist.barrier()
    import axial_vs_amd as ax
    from axial_vs_amd import _lib
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import axvs_oracle as orc
    for kv in args.opt:
        k, v = kv.split("=")
        _lib.check(_lib.lib().axvs_set_option(k.encode(), int(v)), "axvs_set_option")

    B, T, C, H, W = (int(v) for v in args.shape.split(","))
    F = args.d_ffn
    heads = 8
    # synthetic workload (SURVEY.md 8d recipe); every rank gets its own clip
    w = orc.random_weights(orc.axial_layer_param_shapes(C, F), 0)
    src_cpu, pos_cpu = orc.synthetic_clip(B, T, C, H, W, seed=rank)
    layer = ax.TemporalAxialTrajectoryAttentionLayer(C, F, n_heads=heads, mfma_dtype=args.dtype).eval()
    layer.load_state_dict(w, strict=True)
    layer = layer.to(dev)
    src = src_cpu.to(dev)
    # `pos` as the reference's callers make it (WC/msdeformattn.py:108-115): PositionEmbeddingSine3D on the device.  The tensor
    # carries its specification, so the layer evaluates the embedding in the q/k loaders instead of reading 16.8 MB per pass;
    # --tensor-pos passes an untagged copy (read from HBM like any tensor).
    pos = ax.PositionEmbeddingSine3D(C // 2, normalize=True).channels_last(B, T, H, W, dev)
    assert float((pos.cpu() - pos_cpu).abs().max()) < 1e-5
    if args.tensor_pos:
        pos = pos.clone()

    gathered = None
    side = None
    if args.gather and world > 1:
        gathered = torch.empty(world * B * T, H * W, C, device=dev)
        side = torch.cuda.Stream(dev)

    # one step = one layer forward on the resident clip.  --graph replays the 4 launches from a captured HIP graph
    # (axial_vs_amd.GraphedForward); measured on MI355X / ROCm 7.2 it is NOT faster at this size (137.7 vs 133.5 us per step: the
    # Python launch path already keeps ahead of 130 us of GPU work and a graph launch costs more than 4 kernel launches), so the
    # default stays the plain path.
    graphed = ax.GraphedForward(layer, src, pos) if args.graph else None

    def step():
        out = graphed()[0] if graphed is not None else layer(src, pos)[0]
        if gathered is not None:
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                dist.all_gather_into_tensor(gathered, out)
                out.record_stream(side)
        return out

    for _ in range(max(args.warmup, 1)):
        out = step()
    torch.cuda.synchronize(dev)

    # ---- timed region: exactly K steps between barrier + synchronize on both sides ----
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    assert torch.isfinite(out).all()

    frames = world * B * T * args.steps
    value = frames / elapsed
    ms_per_step = elapsed / args.steps * 1e3

    result = {
        "metric": "axial-trajectory-attn fwd frames/sec at [B=1,T=4,C=256,H=W=64]; 1/2/4/8 GPU",
        "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"TemporalAxialTrajectoryAttentionLayer fwd, [B={B},T={T},C={C},H={H},W={W}] per GPU, "
                               f"heads={heads}, d_ffn={F}, fp32 in/out, {args.dtype} MFMA operands",
                   "shape_per_gpu": [B, T, C, H, W], "pos": "tensor read from HBM" if args.tensor_pos else "PositionEmbeddingSine3D, evaluated in-kernel",
                   "launch": "python per step" if graphed is None else "hipGraph replay (4 kernels)",
                   "parallelism": f"dp{world} (clips sharded over ranks"
                   + (", RCCL all-gather of outputs overlapped)" if gathered is not None else ", no collective)")},
    }

    if rank == 0:
        # ---- roofline: HIP events on the launch stream (torch's current stream), averaged over the same K steps ----
        L = _lib.lib()
        hip = ctypes.CDLL("libamdhip64.so")
        hip.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
        nst = L.axvs_profile_stages(None, 0)
        evs = (ctypes.c_void_p * nst)()
        for i in range(nst):
            e = ctypes.c_void_p()
            assert hip.hipEventCreate(ctypes.byref(e)) == 0
            evs[i] = e.value
        stage_ms = [0.0] * nst
        reps = min(args.steps, 50)
        L.axvs_profile_stages(evs, nst)
        for _ in range(reps):
            layer(src, pos)
            torch.cuda.synchronize(dev)
            for i in range(1, L.axvs_profile_stage_count()):
                ms = ctypes.c_float()
                assert hip.hipEventElapsedTime(ctypes.byref(ms), evs[i - 1], evs[i]) == 0
                stage_ms[i] += ms.value / reps
        L.axvs_profile_stages(None, 0)
        # whole-forward duration with events bracketing K back-to-back forwards (no host sync inside)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.steps):
            step()
        e1.record()
        torch.cuda.synchronize(dev)
        fwd_ms = e0.elapsed_time(e1) / args.steps
        flops = layer_flops(B, T, H, W, C, F)
        achieved = flops / (fwd_ms * 1e-3) / 1e12
        nrun = L.axvs_profile_stage_count()
        names = [L.axvs_profile_stage_name(i).decode() for i in range(nrun)]
        kernels = {names[i]: round(stage_ms[i] * 1e3, 2) for i in range(1, nrun)}
        dom = max(kernels, key=kernels.get)
        # algorithmic FLOPs per launch (SURVEY 8d terms) -> per-kernel fraction of the MFMA peak, from the same HIP events
        def f_qkv(S, L):
            return S * T * L * C * C * 6
        def f_traj(S, L):
            N = T * L
            return S * (N * C * C * (4 + 4 * T) + 4 * N * N * C + 4 * N * T * C)
        f_ffn = 4 * B * T * H * W * C * F
        stage_flops = {"h.qkv_proj": f_qkv(B * W, H), "w.qkv_proj": f_qkv(B * H, W), "h.traj_fused": f_traj(B * W, H),
                       "w.traj_fused": f_traj(B * H, W), "w.traj_fused+ffn": f_traj(B * H, W) + f_ffn, "norm1+ffn+norm2": f_ffn}
        stage_frac = {k: round(stage_flops[k] / (v * 1e-6) / 1e12 / MFMA_PEAK_TFLOPS, 4) for k, v in kernels.items()
                      if k in stage_flops and v > 0}
        # HBM traffic per layer forward from the TCC counters: collected by tools/pmc_traffic.sh (rocprofv3 --pmc passes of this
        # very command cannot run inside the timed process); reported only for the workload it was measured on
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic_pmc.json")
        if os.path.exists(tpath) and (B, T, C, H, W, F, args.dtype) == (1, 4, 256, 64, 64, 1024, "f16") and not args.opt:
            tj = json.load(open(tpath))
            traffic, traffic_src = int(tj["layer_total_MB"] * 1e6), "profiles/hbm_traffic_pmc.json (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE)"
        result["roofline"] = {
            "bound": "mfma", "achieved": round(achieved, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / MFMA_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_unit": "bytes per layer forward",
            "traffic_source": traffic_src,
            "kernel": "axial layer forward (all launches of one step)", "launch_us": round(fwd_ms * 1e3, 2),
            "algorithmic_gflop": round(flops / 1e9, 2), "algorithmic_mbytes": round(layer_bytes(B, T, H, W, C, F) / 1e6, 2),
            "hbm_frac_if_memory_bound": round(layer_bytes(B, T, H, W, C, F) / (fwd_ms * 1e-3) / 8e12, 4),
            "stage_us": kernels, "stage_frac": stage_frac, "dominant_stage": dom,
        }

        # ---- CPU baseline: the oracle (a torch CPU port of the reference) on this host, same workload ----
        if not args.no_cpu_baseline and world == 1:      # reported at N = 1 only (the other ranks would just wait)
            ncpu = os.cpu_count() or 1
            fwd = lambda: orc.axial_layer(src_cpu, pos_cpu, w, heads, want_attn=False)
            with torch.no_grad():
                # torch's CPU kernels stop scaling (and thrash) far below a big host's core count: probe a few
                # thread counts with one forward each and time the best one
                best = None
                for nt in sorted({min(ncpu, c) for c in (8, 16, 32, 64)}):
                    torch.set_num_threads(nt)
                    fwd()
                    t1 = time.perf_counter()
                    fwd()
                    dt = time.perf_counter() - t1
                    if best is None or dt < best[1]:
                        best = (nt, dt)
                cores = best[0]
                torch.set_num_threads(cores)
                times = []
                budget = time.perf_counter() + args.cpu_seconds
                while len(times) < 3 or (time.perf_counter() < budget and len(times) < 50):
                    t1 = time.perf_counter()
                    fwd()
                    times.append(time.perf_counter() - t1)
            med = statistics.median(times)
            result["cpu_baseline"] = {
                "value": round(B * T / med, 2), "unit": "frames/s", "cores": cores, "kind": "port",
                "sample": f"{len(times)} forwards of the same [B={B},T={T},C={C},H={H},W={W}] layer after 1 warm-up, "
                          f"fp32, torch CPU {cores} threads (best of 8/16/32/64 on {ncpu} logical CPUs), median {med * 1e3:.1f} ms"}
        print(json.dumps(result), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
