"""128-row FFN tiles (ffn_wide_kernel) against the 64-row kernel through axvs_ffn_fwd: bit-compare and time, per row count.
   tools/ffn_wide_check.py [M ...]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
import torch
import axvs_oracle as orc
import axial_vs_amd as ax
from axial_vs_amd import _lib
C = 256
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
for F in (1024, 512, 2048):
    w = orc.random_weights(orc.axial_layer_param_shapes(C, F), 3)
    layer = ax.TemporalAxialTrajectoryAttentionLayer(C, F, n_heads=8).eval()
    layer.load_state_dict(w, strict=True)
    layer = layer.cuda()
    packed = layer._pack()
    for M in [int(a) for a in sys.argv[1:]] or [21504, 16384, 32768, 65536, 20000, 129, 8192 + 37]:
        xs = torch.randn(M, C, device="cuda") * 1.7 + 0.3
        outs, times = [], []
        ws = torch.empty(L.axvs_ffn_workspace_bytes(M, C, F), dtype=torch.uint8, device="cuda")
        for mode in (4, 2):                       # plan_force: 4 = never the 128-row tiles, 2 = always
            _lib.check(L.axvs_set_option(b"plan_force", mode), "opt")
            out = torch.full_like(xs, float("nan"))
            def run(): _lib.check(L.axvs_ffn_fwd(xs.data_ptr(), out.data_ptr(), packed.data_ptr(), M, C, 8, F, 0, ws.data_ptr(), ws.numel(), st), "ffn")
            for _ in range(20): run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200): run()
            e1.record(); torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1) * 5)
            outs.append(out.clone())
        _lib.check(L.axvs_set_option(b"plan_force", 0), "opt")
        same = torch.equal(outs[0], outs[1])
        print(f"F={F} M={M:6d}: 64-row {times[0]:7.2f} us   128-row {times[1]:7.2f} us (both incl. the input copy)   bit-identical: {same}"
              + ("" if same else f"  max diff {float((outs[0]-outs[1]).abs().max()):.3e} nan {int(torch.isnan(outs[1]).sum())}"))
