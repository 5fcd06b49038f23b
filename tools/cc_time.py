"""Time the cross-clip module at BASELINE config 4 (tuning helper)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
import torch
import axvs_oracle as orc
import axial_vs_amd as ax
from axial_vs_amd import _lib
mod = ax.CrossClipTrackingModule(num_layers=4, num_classes=124, attn_drop=0.0, aspp_drop=0.0, kernel_sizes=[3, 3, 3],
                                 atrous_rates=[1, 2, 3], norm_fn="ln", num_clip_frames=4).eval()
shapes = {k: tuple(v.shape) for k, v in mod.state_dict().items() if v.dtype.is_floating_point}
sd = mod.state_dict(); sd.update(orc.random_weights(shapes, 0)); mod.load_state_dict(sd)
mod = mod.cuda(); mod.eval_outputs_on_cpu = False
cq = torch.randn(1, 128, 4, 256, device="cuda")
pf = torch.nn.functional.normalize(torch.randn(1, 128, 16, 64, 64, device="cuda"), dim=1)
for kv in sys.argv[1:]:          # library options: key=value (axvs_set_option)
    k, v = kv.split("=")
    _lib.check(_lib.lib().axvs_set_option(k.encode(), int(v)), k)
for _ in range(5): mod(cq, pf)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): mod(cq, pf)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 50
print(f"cross-clip module, cfg 4 (4 layers, Q=128, 4 clips x 4 frames, 64x64): {ms*1e3:.1f} us per forward -> {16/ms*1e3:.0f} frames/s; "
      f"mask output {4*33.5:.0f} MB -> {4*33.5e6/(ms*1e-3)/1e12:.2f} TB/s of writes")
# the same forward replayed from a captured HIP graph (launch-bound module: ~60 small kernels)
g = ax.GraphedForward(mod, cq, pf)
ref = mod(cq, pf)
out = g()
assert torch.equal(out["pred_masks"], ref["pred_masks"]) and torch.equal(out["pred_logits"], ref["pred_logits"])
for _ in range(5): g()
e0.record()
for _ in range(50): g()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 50
print(f"  HIP-graph replay: {ms*1e3:.1f} us per forward (bit-identical outputs)")
import time
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): mod(cq, pf)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"  host enqueue time: {(t1 - t0) / 10 * 1e6:.0f} us per forward")
