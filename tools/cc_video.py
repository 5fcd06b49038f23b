"""Whole-video inference of the cross-clip module: Tc = number of clips of the video (the reference's eval path feeds all clips
at once, maxtron_cc_model.py:271-309).  us per forward for growing Tc, and the per-kernel split at the largest one."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
import torch
import axvs_oracle as orc
import axial_vs_amd as ax
mod = ax.CrossClipTrackingModule(num_layers=4, num_classes=124, attn_drop=0.0, aspp_drop=0.0, kernel_sizes=[3, 3, 3],
                                 atrous_rates=[1, 2, 3], norm_fn="ln", num_clip_frames=2).eval()
shapes = {k: tuple(v.shape) for k, v in mod.state_dict().items() if v.dtype.is_floating_point}
sd = mod.state_dict(); sd.update(orc.random_weights(shapes, 0)); mod.load_state_dict(sd)
mod = mod.cuda(); mod.eval_outputs_on_cpu = False
for Tc in [4, 8, 10, 12, 16, 24, 40]:
    cq = torch.randn(1, 128, Tc, 256, device="cuda")
    pf = torch.nn.functional.normalize(torch.randn(1, 128, Tc * 2, 64, 64, device="cuda"), dim=1)
    for _ in range(10): mod(cq, pf)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 20
    for _ in range(n): mod(cq, pf)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"Tc={Tc:3d} clips x 2 frames, 64x64: {dt*1e6:8.1f} us per forward ({Tc*2/dt:.0f} frames/s)")
