import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
import torch
import axvs_oracle as orc
import axial_vs_amd as ax
from axial_vs_amd import _lib
M, C, F = int(sys.argv[1]) if len(sys.argv) > 1 else 16384, 256, 1024
w = orc.random_weights(orc.axial_layer_param_shapes(C, F), 3)
layer = ax.TemporalAxialTrajectoryAttentionLayer(C, F, n_heads=8).eval()
layer.load_state_dict(w, strict=True)
layer = layer.cuda()
packed = layer._pack()
L = _lib.lib()
xs = torch.randn(M, C, device="cuda")
out = torch.empty_like(xs)
ws = torch.empty(L.axvs_ffn_workspace_bytes(M, C, F), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def run(): _lib.check(L.axvs_ffn_fwd(xs.data_ptr(), out.data_ptr(), packed.data_ptr(), M, C, 8, F, 0, ws.data_ptr(), ws.numel(), st), "ffn")
for _ in range(10): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100): run()
e1.record(); torch.cuda.synchronize()
print(f"ffn (incl. 16.8MB copy) {e0.elapsed_time(e1) * 10:.2f} us")
