"""Per-stage event timing of one cross-clip layer + heads at cfg 4."""
import sys, os, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
import torch, axvs_oracle as orc, axial_vs_amd as ax
from axial_vs_amd import _lib
mod = ax.CrossClipTrackingModule(num_layers=1, num_classes=124, attn_drop=0.0, aspp_drop=0.0, kernel_sizes=[3, 3, 3], atrous_rates=[1, 2, 3], norm_fn="ln", num_clip_frames=4).eval()
shapes = {k: tuple(v.shape) for k, v in mod.state_dict().items() if v.dtype.is_floating_point}
sd = mod.state_dict(); sd.update(orc.random_weights(shapes, 0)); mod.load_state_dict(sd)
mod = mod.cuda(); mod.eval_outputs_on_cpu = False
cq = torch.randn(1, 128, 4, 256, device="cuda"); pf = torch.nn.functional.normalize(torch.randn(1, 128, 16, 64, 64, device="cuda"), dim=1)
for _ in range(3): mod(cq, pf)
L = _lib.lib(); hip = ctypes.CDLL("libamdhip64.so")
hip.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]; hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
nst = L.axvs_profile_stages(None, 0); evs = (ctypes.c_void_p * nst)()
for i in range(nst):
    e = ctypes.c_void_p(); hip.hipEventCreate(ctypes.byref(e)); evs[i] = e.value
# the module makes two C calls per layer (layer, heads); profile each separately by calling the C-ABI through the module once per kind
import types
acc = {}
orig_layer, orig_heads = L.axvs_cc_layer_fwd, L.axvs_cc_heads_fwd
def wrap(fn):
    def w(*a):
        L.axvs_profile_stages(evs, nst)
        rc = fn(*a); torch.cuda.synchronize()
        for i in range(1, L.axvs_profile_stage_count()):
            ms = ctypes.c_float(); hip.hipEventElapsedTime(ctypes.byref(ms), evs[i - 1], evs[i])
            nm = L.axvs_profile_stage_name(i).decode(); acc[nm] = acc.get(nm, 0) + ms.value
        L.axvs_profile_stages(None, 0)
        return rc
    return w
class P:  # proxy lib
    def __getattr__(s, k): return wrap(getattr(L, k)) if k in ("axvs_cc_layer_fwd", "axvs_cc_heads_fwd") else getattr(L, k)
_lib_lib = _lib.lib
_lib.lib = lambda: P()
for _ in range(10): mod(cq, pf)
print("  ".join(f"{k}={v / 10 * 1e3:.1f}us" for k, v in acc.items()))
