"""Do the merged forms on small tiles hold with COLD weights?  A stack of 6 axial layers with their own weights (TemporalEncoder), timed per forward (events around the
forward only) warm (back to back) and cold (a 1 GiB copy between forwards: L2 and MALL hold none of the weights), option off / on, two rounds.
    python3 tools/r5/stack_ab.py merge_small 1,4,16,16 1,4,32,32 ...   |   python3 tools/r5/stack_ab.py merge_mid 1,5,24,40 ..."""
import sys, os
R = os.path.join(os.path.dirname(__file__), "..", "..")
sys.path.insert(0, R)
import torch
import axial_vs_amd as ax
from axial_vs_amd import _lib
L = _lib.lib()
opt = sys.argv[1].encode()
big_a = torch.empty(1 << 28, device="cuda"); big_b = torch.empty(1 << 28, device="cuda")
for sh in sys.argv[2:]:
    B, T, H, W = [int(v) for v in sh.split(",")]
    enc = ax.TemporalEncoder(256, 1024, 0.0, 0.0, "relu", 8, "axial-trajectory", 6).eval().cuda()
    s = torch.randn(B * T, H * W, 256, device="cuda")
    p = ax.PositionEmbeddingSine3D(128, normalize=True).channels_last(B, T, H, W, "cuda")
    row = []
    for val in (0, 1, 0, 1):
        _lib.check(L.axvs_set_option(opt, val), "axvs_set_option")
        for _ in range(10): enc(s, p)
        torch.cuda.synchronize()
        res = {}
        for cold in (False, True):
            ts = []
            for _ in range(40):
                if cold:
                    big_b.copy_(big_a)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); enc(s, p); e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            ts.sort()
            res[cold] = ts[len(ts) // 4]
        row.append((val, res[False], res[True]))
    L.axvs_set_option(opt, 1 if opt == b"merge_mid" else 0)
    print(f"{sh:>10s} 6-layer stack, us per forward (warm | cold):  " + "   ".join(f"{opt.decode()}={v}: {a:7.1f} | {c:7.1f}" for v, a, c in row), flush=True)
