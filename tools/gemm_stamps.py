"""Phase stamps of the training tier's X W^T GEMM (tools/diag_stamps_tr.so, -DAXVS_STAMPS_TR): the LAST tr_gemm_nt_kernel launch of a
forward of a small layer -- linear2, [M x 256 x d_ffn] -- workgroups 0..7, all 8 waves.
    AXVS_LIB_PATH=tools/diag_stamps_tr.so python tools/gemm_stamps.py [H W F]"""
import os, sys, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
import torch, numpy as np
import axvs_oracle as orc
import axial_vs_amd as ax
from axial_vs_amd import _lib
H, W, F = [int(a) for a in sys.argv[1:4]] if len(sys.argv) > 3 else (16, 16, 256)
B, T, C = 1, 4, 256
w = orc.random_weights(orc.axial_layer_param_shapes(C, F), 0)
src, pos = orc.synthetic_clip(B, T, C, H, W, 0)
layer = ax.TemporalAxialTrajectoryAttentionLayer(C, F, dropout=0.0, attn_drop=0.0, n_heads=8)
layer.load_state_dict(w, strict=True)
layer = layer.cuda().train()
s, p = src.cuda(), pos.cuda()
raw = ctypes.CDLL(_lib.LIB_PATH)
names = ["launch -> operands of k-step 0 staged (prologue)", "k-loop", "accumulators -> LDS + barrier", "row stores issued (epilogue)", "stores complete"]
for rep in range(3):
    with torch.no_grad():
        layer(s, p)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (64 * 64))()
    raw.axvs_debug_read_stamps_tr(buf, 64 * 64)
    a = np.array(buf, dtype=np.uint64).reshape(64, 64).astype(np.int64)
    M, N, K, NS, flags = [int(x) for x in a[6][:5]]
    nwg = min(8, (M + 127) // 128)
    st = a[:6, :nwg * 8]
    d = np.diff(st, axis=0)
print(f"last tr_gemm_nt_kernel<{NS}> launch: M={M} N={N} K={K} (flags res/bias/drop = {flags}), {((M + 127) // 128) * ((N + 127) // 128)} workgroups; cycles, median over {nwg * 8} waves")
for i, nm in enumerate(names):
    print(f"  {nm:52s} {int(np.median(d[i])):7d}   (min {int(d[i].min())}, max {int(d[i].max())})")
print(f"  total {int(np.median(st[5] - st[0]))} cycles")
ks = a[10:14, :nwg * 8]                         # one k-step (the third): entry, MFMAs issued, next tile stored, barrier passed
dk = np.diff(ks, axis=0)
for i, nm in enumerate(["k-step 2: loads requested, fragments read, MFMAs issued", "          next tile: global loads waited for, split, stored to LDS", "          barrier"]):
    print(f"  {nm:70s} {int(np.median(dk[i])):6d}   (min {int(dk[i].min())}, max {int(dk[i].max())})")
