"""Stage timing of MSDeformAttn.forward at the cfg-3 size: N = B*T = 4 frames, levels 64x64 + 32x32 + 16x16, C = 256."""
import sys, os, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import axial_vs_amd as ax
from axial_vs_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
shapes = [(64, 64), (32, 32), (16, 16)]
S = sum(h * w for h, w in shapes)
torch.manual_seed(0)
mod = ax.MSDeformAttn(256, 3, 8, 4).eval()
with torch.no_grad():
    mod.sampling_offsets.weight.uniform_(-0.1, 0.1)
    mod.attention_weights.weight.uniform_(-0.25, 0.25)
mod = mod.cuda()
src = torch.randn(N, S, 256, device="cuda")
q = src + 0.5 * torch.randn(N, S, 256, device="cuda")
refs = []
for (h, w) in shapes:
    ys, xs = torch.meshgrid(torch.linspace(0.5, h - 0.5, h) / h, torch.linspace(0.5, w - 0.5, w) / w, indexing="ij")
    refs.append(torch.stack([xs.reshape(-1), ys.reshape(-1)], -1))
ref = torch.cat(refs, 0)[None, :, None, :].expand(N, S, 3, 2).contiguous().cuda()
for _ in range(5): mod(q, ref, src, shapes)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
K = 50
e0.record()
for _ in range(K): mod(q, ref, src, shapes)
e1.record(); torch.cuda.synchronize()
print(f"MSDeformAttn fwd N={N} S={S}: {e0.elapsed_time(e1) / K * 1e3:.1f} us")
L = _lib.lib()
hip = ctypes.CDLL("libamdhip64.so")
hip.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
nst = L.axvs_profile_stages(None, 0)
evs = (ctypes.c_void_p * nst)()
for i in range(nst):
    e = ctypes.c_void_p(); hip.hipEventCreate(ctypes.byref(e)); evs[i] = e.value
L.axvs_profile_stages(evs, nst)
acc = {}
for _ in range(20):
    mod(q, ref, src, shapes); torch.cuda.synchronize()
    for i in range(1, L.axvs_profile_stage_count()):
        ms = ctypes.c_float(); hip.hipEventElapsedTime(ctypes.byref(ms), evs[i - 1], evs[i])
        nm = L.axvs_profile_stage_name(i).decode(); acc[nm] = acc.get(nm, 0) + ms.value / 20
L.axvs_profile_stages(None, 0)
print("  ".join(f"{k}={v * 1e3:.1f}us" for k, v in acc.items()))
