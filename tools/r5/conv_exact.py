import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from axial_vs_amd import _lib
L = _lib.lib()
for (N, HW, Cin, Cout) in [(2, 1075, 2048, 256), (4, 256, 768, 256), (4, 1024, 384, 256), (5, 960, 1024, 256), (5, 240, 2048, 256), (2, 16393, 512, 256), (2, 4165, 1024, 256), (2, 16393, 256, 512), (4, 4096, 192, 256)]:
    g = torch.Generator().manual_seed(N * 1000 + HW)
    x = torch.randn(N, Cin, HW, generator=g)
    w = torch.randn(Cout, Cin, generator=g) / Cin ** 0.5
    b, gw, gb = torch.randn(Cout, generator=g) * 0.1, 1 + 0.1 * torch.randn(Cout, generator=g), 0.1 * torch.randn(Cout, generator=g)
    ref = torch.nn.functional.group_norm(torch.einsum("oc,ncp->nop", w.double(), x.double()) + b.double()[None, :, None], 32, gw.double(), gb.double(), 1e-5)
    dw, db, dgw, dgb, dx = (t_.cuda().contiguous() for t_ in (w, b, gw, gb, x))
    ps = _lib.AxvsConvGnParams(dw.data_ptr(), db.data_ptr(), dgw.data_ptr(), dgb.data_ptr())
    packed = torch.empty(L.axvs_conv1x1_gn_packed_bytes(Cin, Cout), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(L.axvs_conv1x1_gn_pack(C.byref(ps), packed.data_ptr(), Cin, Cout, 0, st), "pack")
    ws = torch.empty(L.axvs_conv1x1_gn_workspace_bytes(N, HW, max(Cin, Cout), 32), dtype=torch.uint8, device="cuda")
    rows = torch.empty(N, HW, Cout, device="cuda")
    for ex in (1024, 384, 1 << 20):
        L.axvs_set_option(b"conv_nt128_splitk", ex)
        def run():
            _lib.check(L.axvs_conv1x1_gn_fwd(dx.data_ptr(), 0, 0, 0, rows.data_ptr(), 1, HW * Cout, Cout, packed.data_ptr(), N, HW, Cin, Cout, 32, 1e-5, 0, ws.data_ptr(), ws.numel(), st), "fwd")
        for _ in range(5): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): run()
        e1.record(); torch.cuda.synchronize()
        err = float((rows.cpu().permute(0, 2, 1).double() - ref).abs().max() / ref.abs().max())
        print(f"N={N} HW={HW} {Cin}->{Cout} splitk={ex}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per conv+GN, max/max {err:.2e}", flush=True)
    L.axvs_set_option(b"conv_nt128_splitk", 1024)
