"""Standalone FFN-kernel check: repeat axvs_ffn_fwd on fixed input, compare with a float64 CPU reference."""
import sys, os, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
import torch
import axvs_oracle as orc
import axial_vs_amd as ax
from axial_vs_amd import _lib
M, C, F = int(sys.argv[1]) if len(sys.argv) > 1 else 16384, 256, 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
w = orc.random_weights(orc.axial_layer_param_shapes(C, F), 3)
layer = ax.TemporalAxialTrajectoryAttentionLayer(C, F, n_heads=8).eval()
layer.load_state_dict(w, strict=True)
layer = layer.cuda()
packed = layer._pack()
g = torch.Generator().manual_seed(5)
x = torch.randn(M, C, generator=g) * 2 + 0.3
xd = x.double()
y = orc._layer_norm(xd, w, "norm1")
ref = orc._layer_norm(y + orc._linear(torch.relu(orc._linear(y, w, "linear1")), w, "linear2"), w, "norm2")
L = _lib.lib()
xs = x.cuda()
ws = torch.empty(L.axvs_ffn_workspace_bytes(M, C, F), dtype=torch.uint8, device="cuda")
DBG = os.environ.get("AXVS_DBG2")
if DBG:
    raw = ctypes.CDLL(_lib.LIB_PATH); raw.axvs_debug_buffer.argtypes = [ctypes.c_void_p]
    dbg = torch.zeros(256 * 8 * 64 * 8, device="cuda"); raw.axvs_debug_buffer(dbg.data_ptr())
    L.axvs_set_option(b"generic_only", 0)
    runs = []
    for it in range(reps):
        dbg.zero_(); out = torch.empty_like(xs)
        _lib.check(L.axvs_ffn_fwd(xs.data_ptr(), out.data_ptr(), packed.data_ptr(), M, C, 8, F, 0, ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream), "ffn")
        torch.cuda.synchronize()
        e = (out.cpu().double() - ref).abs().max().item()
        runs.append((e, dbg.clone().reshape(256, 8, 64, 8)))
    good = [d for e, d in runs if e < 5e-3][0]
    names = ["rs(partial row sum)", "stats.mu", "stats.rstd", "sum over waves", "sumsq over waves", "rq partial"]
    for i, (e, d) in enumerate(runs):
        if e < 5e-3: continue
        idx = (d != good).nonzero()
        print(f"it {i}: err {e:.2e}, dbg diffs {idx.shape[0]}")
        seen = set()
        for (wg, wv, row, k) in idx.tolist():
            key = (wg, wv, k)
            if key in seen: continue
            seen.add(key)
            if len(seen) > 30: break
            print(f"   wg {wg} wave {wv} row {row}: {names[k]}: {d[wg,wv,row,k].item():.6f} vs good {good[wg,wv,row,k].item():.6f}")
    sys.exit(0)
if os.environ.get("AXVS_DET"):
    # determinism only: count outputs that differ from the most frequent one
    L.axvs_set_option(b"generic_only", 0)
    outs = []
    for it in range(reps):
        out = torch.empty_like(xs)
        _lib.check(L.axvs_ffn_fwd(xs.data_ptr(), out.data_ptr(), packed.data_ptr(), M, C, 8, F, 0, ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream), "ffn")
        torch.cuda.synchronize()
        outs.append(out.clone())
    keys = [float(o.double().sum()) for o in outs]
    from collections import Counter
    mode, cnt = Counter(keys).most_common(1)[0]
    print(f"deterministic runs {cnt}/{reps}; err of mode vs ref {(outs[keys.index(mode)].cpu().double()-ref).abs().max().item():.2e}")
    sys.exit(0)
for gen in (1, 0):
    L.axvs_set_option(b"generic_only", gen)
    res = []
    for it in range(reps):
        out = torch.empty_like(xs)
        _lib.check(L.axvs_ffn_fwd(xs.data_ptr(), out.data_ptr(), packed.data_ptr(), M, C, 8, F, 0, ws.data_ptr(), ws.numel(),
                                  torch.cuda.current_stream().cuda_stream), "ffn")
        torch.cuda.synchronize()
        e = (out.cpu().double() - ref).abs()
        bad = (e > 0.02).any(1).nonzero().flatten()
        res.append(f"{e.max().item():.1e}" + (f"[wg{torch.unique(bad//64).tolist()[:4]} r{torch.unique(bad%64).tolist()[:3]}..]" if bad.numel() else ""))
    print("generic" if gen else "fused  ", " ".join(res))
