"""nn.Module mirrors of the reference's within-clip trajectory-attention classes.

Same constructor arguments, attribute / parameter names (state-dict keys) and call signatures as

    WC = MaXTron_Video-kMaX/maxtron_deeplab/modeling/within_clip_tracking_module/temporal_attention.py
    TL = MaXTron_Tube-Link/mmdet/models/plugins/msdeformattn_pixel_decoder.py

so checkpoints load with ``strict=True`` and the callers (WC/msdeformattn.py:262, TL:623-627) are unchanged.
``forward`` hands raw device pointers to libaxvs.so; PyTorch only owns the memory and the stream.

Forward-only: modules must be in ``eval()`` mode (the training path with dropout / autograd is out of scope,
SURVEY.md section 8f).  There is no CPU fallback -- tensors must live on the GPU.
"""
from __future__ import annotations

import ctypes as C
import functools
import math
import operator
import threading
import weakref
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn
from torch import Tensor

from . import _lib

_DEFAULT_DTYPE = "f16"


def set_default_dtype(name: str) -> None:
    """MFMA operand type used by modules that were not given an explicit ``mfma_dtype``: 'f16' | 'bf16' | 'f32' (the last one
    only exists for TemporalAxialTrajectoryAttentionLayer / TemporalEncoder: see TemporalAxialTrajectoryAttentionLayer.forward)."""
    global _DEFAULT_DTYPE
    if name not in _lib.DTYPES and name != "f32":
        raise ValueError(f"unknown dtype {name!r}")
    _DEFAULT_DTYPE = name


def default_operand_dtype() -> str:
    """16-bit operand type of the modules that have no fp32 tier (everything but the axial layer / TemporalEncoder): the default,
    or 'f16' when the default is 'f32'."""
    return "f16" if _DEFAULT_DTYPE == "f32" else _DEFAULT_DTYPE


# ---------------------------------------------------------------------------------------------
# plumbing
# ---------------------------------------------------------------------------------------------
_workspaces: Dict[Tuple[str, int], Tensor] = {}
_sync_tls = threading.local()       # per calling thread: which sync words / status word the library holds, graph-construction overrides


def _workspace(device: torch.device, nbytes: int, stream_handle: Optional[int] = None) -> Tensor:
    """Grow-only scratch buffer per (device, stream) from torch's caching allocator.  `stream_handle`: the current stream's handle
    when the caller has it already (torch.cuda.current_stream costs ~4 us per call)."""
    key = (str(device), stream_handle if stream_handle is not None else _stream(device))
    ov = getattr(_sync_tls, "override", None)
    store = _workspaces if ov is None else ov["ws"]      # a graph under construction owns its scratch (GraphedForward): graphs captured on
    buf = store.get(key)                                 # torch's shared capture stream must not bake one common workspace in
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=device)
        store[key] = buf
    return buf


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(device: torch.device) -> int:
    """Handle of torch's current stream on `device`.  torch.cuda.current_stream builds a Stream object (~3 us, and a forward of the
    within-clip module asks ~36 times); the raw getter behind it returns the handle itself."""
    if _raw_stream is not None:
        return _raw_stream(device.index if device.index is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(device).cuda_stream


# Arrival counters of the one-launch-per-pass kernels (include/axvs.h, axvs_set_sync_buffer): zero when registered, left zero by
# every launch.  One set per (device, stream) -- calls on different streams may run concurrently (pixel_decoder runs two levels
# side by side) -- registered with the library per calling thread, re-registered only when the (device, stream) changes.
_SYNC_WORDS = 16384
_sync_buffers: Dict[Tuple[int, int], Tensor] = {}


def _select_sync_words(device: torch.device, stream_handle: Optional[int] = None) -> None:
    idx = device.index if device.index is not None else torch.cuda.current_device()
    override = getattr(_sync_tls, "override", None)
    if override is not None:             # GraphedForward: the graph being warmed up / captured owns its counters (see there)
        sh = stream_handle if stream_handle is not None else _stream(device)
        key = ("graph", idx, sh, id(override))
        if getattr(_sync_tls, "key", None) == key:
            return
        buf = override["map"].get((idx, sh))
        if buf is None and override["pool"]:         # one pre-zeroed set per stream the module uses (pixel_decoder: two)
            buf = override["map"][(idx, sh)] = override["pool"].pop()
        if buf is None:                              # more streams than sets: this one runs two launches per pass (same bits)
            _lib.check(_lib.lib().axvs_set_sync_buffer(None, 0), "axvs_set_sync_buffer")
        else:
            _lib.check(_lib.lib().axvs_set_sync_buffer(buf.data_ptr(), buf.numel()), "axvs_set_sync_buffer")
        _sync_tls.key = key
        return
    key = (idx, stream_handle if stream_handle is not None else _stream(device))
    if getattr(_sync_tls, "key", None) == key:
        return
    buf = _sync_buffers.get(key)
    if buf is None:
        if torch.cuda.is_current_stream_capturing():
            # A first use of this stream INSIDE a graph capture (a caller's own torch.cuda.graph around a module): an allocation here
            # would land in the graph's private pool and its zero-fill would become a graph node instead of running now.  No sync words
            # for this capture: the passes run as two launches (same bits).  GraphedForward allocates its words before capturing.
            _lib.check(_lib.lib().axvs_set_sync_buffer(None, 0), "axvs_set_sync_buffer")
            _sync_tls.key = ("capture-without-words", idx)
            return
        buf = _sync_buffers[key] = torch.zeros(_SYNC_WORDS, dtype=torch.int32, device=torch.device("cuda", idx))
    _lib.check(_lib.lib().axvs_set_sync_buffer(buf.data_ptr(), _SYNC_WORDS), "axvs_set_sync_buffer")
    _sync_tls.key = key


def _dev_f32(t: Tensor, what: str) -> Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"axial_vs_amd: {what} must be a GPU tensor (got {t.device}); there is no CPU fallback")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _require_eval(m: nn.Module) -> None:
    if m.training:
        raise NotImplementedError("axial_vs_amd: this module has a forward-only HIP path -- call .eval() (the training tier covers "
                                  "TemporalAxialTrajectoryAttentionLayer / TemporalEncoder('axial-trajectory'), axial_vs_amd/training.py)")


def _ptr(t: Optional[Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


_tensor_data_ptr = torch.Tensor.data_ptr
_tensor_version = operator.attrgetter("_version")


def _param_key(mod: nn.Module, dtype: str):
    """Cheap change detector for the packed-weight caches: (storage pointer, version counter) of every parameter.
    Walking `mod.modules()` / `mod.parameters()` costs ~75 us for a layer (more than half of the GPU time of a forward), so the
    walk is cached as references INTO the owners' dictionaries: (owner._parameters, name) pairs are re-read on every call --
    a Parameter replaced by assignment (`lin.bias = nn.Parameter(...)`, parametrizations, `.to()` with
    overwrite_module_params_on_conversion) is seen at once -- and (parent._modules, name, id(child)) triples detect a replaced
    submodule, which triggers a fresh walk."""
    d = mod.__dict__
    cache = d.get("_axvs_refs")
    if cache is not None:
        for md, name, ident in cache[0]:
            if id(md.get(name)) != ident:
                cache = None
                break
    if cache is None:
        mods, prefs = [], []
        for m in mod.modules():
            for name, child in m._modules.items():
                mods.append((m._modules, name, id(child)))
            for name in m._parameters:
                prefs.append((m._parameters, name))
        cache = (mods, prefs)
        d["_axvs_refs"] = cache
    ps = [pd.get(name) for pd, name in cache[1]]
    try:            # the two C-level sweeps cost a third of a Python loop over the parameters (14 such keys per forward of the within-clip module)
        return (dtype, tuple(map(_tensor_data_ptr, ps)), tuple(map(_tensor_version, ps)))
    except (TypeError, AttributeError):          # a parameter slot holding None (bias=False)
        return (dtype, tuple([None if p is None else (p.data_ptr(), p._version) for p in ps]))


def _has_hooks(mod: nn.Module) -> bool:
    """Forward (pre-)hooks registered on `mod` or globally: callers that bypass `forward()` for a faster data flow check this first."""
    import torch.nn.modules.module as _m
    return bool(mod._forward_hooks or mod._forward_pre_hooks or _m._global_forward_hooks or _m._global_forward_pre_hooks)


def invalidate_pack(mod: nn.Module) -> None:
    """Forget cached parameter references / packed weights of `mod` and its children."""
    for m in mod.modules():
        m.__dict__.pop("_axvs_refs", None)
        if "_packed" in m.__dict__:
            m.__dict__["_packed"] = None


class _on:
    """Make `device` current around a library call when it is not already (the library launches on the current HIP device)."""
    __slots__ = ("idx", "prev")

    def __init__(self, device: torch.device):
        self.idx = device.index if device.index is not None else torch.cuda.current_device()

    def __enter__(self):
        self.prev = torch.cuda.current_device()
        if self.prev != self.idx:
            torch.cuda.set_device(self.idx)

    def __exit__(self, *exc):
        if self.prev != self.idx:
            torch.cuda.set_device(self.prev)
        return False


def _guarded(fn):
    """Run `fn` with the device of its first CUDA tensor argument current (libaxvs launches on the current HIP device and its
    streams / per-device kernel attributes belong to it): tensors on cuda:1 while cuda:0 is current must not launch on 0."""
    @functools.wraps(fn)
    def wrapper(*args, **kw):
        act = getattr(args[0], "activation", "relu") if args and isinstance(args[0], nn.Module) else "relu"
        if act != "relu":           # the layer's FFN activation travels as a thread-local option around the library calls
            with _ffn_activation(act):
                return guarded(*args, **kw)
        return guarded(*args, **kw)

    def guarded(*args, **kw):
        for a in args:
            if isinstance(a, Tensor) and a.is_cuda:
                with _on(a.device):
                    _select_status_word(a.device)
                    if _handoff_policy[0] != "verify" or getattr(_sync_tls, "depth", 0) or torch.cuda.is_current_stream_capturing():
                        return fn(*args, **kw)
                    return _verified(fn, a.device, args, kw)
        return fn(*args, **kw)
    return wrapper


# ---- hand-off policy (round 6) -------------------------------------------------------------------------------------------------------
# "async" (default): library calls never synchronise.  A merged launch whose hand-off wait ran out writes NaN rows (never numbers computed
#     from stale K / V^T), sets the status bit, and the NEXT call on this thread raises after the handler has put the counters back in order.
# "verify": the outermost module call waits for its own stream, reads the status word, and -- if a wait ran out -- zeroes the counters and runs
#     the SAME call again with two launches per pass (bit-identical to the merged form by construction, tests/test_hip_parity.py): the caller gets
#     correct bits and no exception.  Costs the host's run-ahead (one stream synchronisation per forward); meant for GPUs shared with other tenants.
_handoff_policy: List[str] = ["async"]


def set_handoff_policy(policy: str) -> str:
    """'async' | 'verify' (see above).  Returns the previous policy."""
    if policy not in ("async", "verify"):
        raise ValueError("axial_vs_amd: handoff policy must be 'async' or 'verify'")
    prev = _handoff_policy[0]
    _handoff_policy[0] = policy
    return prev


def _verified(fn, device: torch.device, args, kw):
    idx = device.index if device.index is not None else torch.cuda.current_device()
    _sync_tls.depth = 1
    try:
        out = fn(*args, **kw)
        torch.cuda.current_stream(device).synchronize()
        if not int(_status_word(idx)[0]) & 4:
            return out
        _on_sync_timeout()                                   # counters zeroed, bit cleared
        _handoff_recoveries[0] += 1
        L = _lib.lib()
        _lib.check(L.axvs_set_option(b"no_merge_qkv", 1), "axvs_set_option")
        try:
            out = fn(*args, **kw)                            # two launches per pass: nothing to wait for
            torch.cuda.current_stream(device).synchronize()
        finally:
            L.axvs_set_option(b"no_merge_qkv", 0)
        return out
    finally:
        _sync_tls.depth = 0


_handoff_recoveries: List[int] = [0]      # forwards re-run by the "verify" policy (tests, monitoring)


class SineTag:
    """Side note carried by a position tensor made by PositionEmbeddingSine3D: the tensor still holds the real values, but a
    layer that receives it can hand the SPECIFICATION to libaxvs (axvs_axial_layer_fwd_sine3d) and skip reading it.  `version`
    pins the tensor's in-place modification counter: a tensor edited after tagging is used as a plain tensor."""
    __slots__ = ("temperature", "normalize", "scale", "level", "shape", "version")

    def __init__(self, temperature, normalize, scale, level, shape, version):
        self.temperature, self.normalize, self.scale, self.level, self.shape, self.version = temperature, normalize, scale, level, shape, version


def tag_sine3d(pos: Tensor, temperature: float, normalize: bool, scale: float, level: Optional[Tensor] = None) -> Tensor:
    pos._axvs_sine3d = SineTag(float(temperature), bool(normalize), float(scale), level, tuple(pos.shape), pos._version)
    return pos


def _sine_tag(pos: Tensor) -> Optional[SineTag]:
    tag = getattr(pos, "_axvs_sine3d", None)
    if tag is None or tag.shape != tuple(pos.shape) or tag.version != pos._version or not pos.is_contiguous():
        return None
    return tag


# ---- the status word: ALWAYS registered (round 5) ----------------------------------------------------------------------------------
# One int32 per device in PINNED HOST memory (device-visible at the same address): the kernels OR condition bits into it only when a
# condition fires (bit 0: an operand left the fp16 range; bit 2: a hand-off wait of a merged launch ran out -- include/axvs.h), so it
# costs nothing on the normal path (profiles/r5_pinned_status_probe.txt), and the host reads it without a copy or a synchronisation.
# Fail loudly: with a host-readable word the library refuses the NEXT axial-layer call on this thread (AXVS_ERR_STATE -> RuntimeError
# here, after `_on_sync_timeout` has put the counters back in order); `check_status()` is the explicit form.
_status_words: Dict[int, Tensor] = {}
# (which device's word the library currently holds is remembered per calling thread, like the library's own pointer: _sync_tls.status_idx)
_range_check_enabled: List[bool] = [False]


def _status_word(idx: int) -> Tensor:
    w = _status_words.get(idx)
    if w is None:
        w = _status_words[idx] = torch.zeros(1, dtype=torch.int32).pin_memory()
    return w


def _select_status_word(device: torch.device) -> None:
    """The library holds a single status pointer per thread: hand it the word of the device the next call launches on."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if getattr(_sync_tls, "status_idx", None) == idx:
        return
    if idx not in _status_words and torch.cuda.is_current_stream_capturing():
        # a device's FIRST library call inside somebody's graph capture: pinning host memory (hipHostMalloc) would invalidate the capture -- no status
        # word for this capture (nothing is registered: merged launches are off without arrival counters too, see _select_sync_words)
        return
    _lib.check(_lib.lib().axvs_set_status_buffer(_status_word(idx).data_ptr()), "axvs_set_status_buffer")
    _sync_tls.status_idx = idx


def _on_sync_timeout() -> None:
    """A merged launch reported AXVS_STATUS_SYNC_TIMEOUT: wait for the devices concerned, zero their arrival counters (the timed-out
    launch left them non-zero), clear the bit.  The caller raises afterwards; the next call starts from a clean state."""
    for idx, w in _status_words.items():
        if int(w[0]) & 4:
            torch.cuda.synchronize(idx)
            for (kidx, _), buf in list(_sync_buffers.items()):
                if kidx == idx:
                    buf.zero_()
            for g in list(_live_graphs):                  # the counters baked into captured graphs (GraphedForward owns its sets)
                for buf in g._sync:
                    if buf.device.index == idx:
                        buf.zero_()
            torch.cuda.synchronize(idx)
            w[0] = int(w[0]) & ~4


_live_graphs: "weakref.WeakSet" = weakref.WeakSet()      # GraphedForward instances (their arrival counters are zeroed by the handler too)
_lib._state_handler = _on_sync_timeout


def check_status(device=None, synchronize: bool = True) -> None:
    """Raise RuntimeError if a merged q/k/v + trajectory launch on `device` (default: every device used so far) gave up waiting for
    its sibling row tiles -- the outputs of that forward are invalid.  `synchronize=True` waits for the launches still in flight
    first; False reads the word as it is (free).  The counters are zeroed and the bit cleared before the error is raised."""
    if device is None:
        idxs = list(_status_words)
    else:
        dev = torch.device(device)
        idxs = [dev.index if dev.index is not None else torch.cuda.current_device()]
    hit = []
    for idx in idxs:
        w = _status_words.get(idx)
        if w is None:
            continue
        if synchronize:
            torch.cuda.synchronize(idx)
        if int(w[0]) & 4:
            hit.append(idx)
    if hit:
        _on_sync_timeout()
        raise RuntimeError(f"axial_vs_amd: a merged q/k/v + trajectory launch on cuda:{hit} gave up waiting for its sibling row tiles "
                           "(AXVS_STATUS_SYNC_TIMEOUT): the outputs of that forward are invalid.  The arrival counters have been "
                           "zeroed; run again, or keep two launches per pass with axvs_set_option('no_merge_qkv', 1)")


def enable_range_check(device="cuda") -> None:
    """The fused q/k/v loaders flag operands outside the fp16 range (the f16 operand mode would turn them into inf silently) in the
    status word -- always registered since round 5; this call only arms `range_check_report()` (kept for the round-1 API)."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    _status_word(idx)
    _range_check_enabled[0] = True


def range_check_report(device="cuda", reset: bool = True) -> bool:
    """True if a fused loader saw an operand beyond the fp16 range since the last reset (synchronises the device).  A hand-off
    timeout recorded in the same word is NOT swallowed: it raises (check_status)."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if not _range_check_enabled[0]:
        raise RuntimeError("axial_vs_amd: call enable_range_check() first")
    torch.cuda.synchronize(idx)
    w = _status_word(idx)
    v = int(w[0])
    if reset:
        w[0] = v & ~1
    if v & 4:
        check_status(dev, synchronize=False)
    return bool(v & 1)


def disable_range_check() -> None:
    """Disarm `range_check_report()`; the status word itself stays registered (it also carries the hand-off timeout bit)."""
    _range_check_enabled[0] = False
    for w in _status_words.values():
        w[0] = int(w[0]) & ~1


def _traj_struct(m: "TrajectoryAttention", keep: list) -> _lib.AxvsTrajParams:
    C_ = m.proj.weight.shape[0]
    if hasattr(m, "qkv"):  # cross-clip flavour: slices of the fused projection
        w, b = _dev_f32(m.qkv.weight.detach(), "qkv.weight"), _dev_f32(m.qkv.bias.detach(), "qkv.bias")
        qw, kw, vw = w[:C_], w[C_:2 * C_], w[2 * C_:]
        qb, kb, vb = b[:C_], b[C_:2 * C_], b[2 * C_:]
    else:
        qw, qb = m.q.weight, m.q.bias
        kw, kb = m.k.weight, m.k.bias
        vw, vb = m.v.weight, m.v.bias
    ts = [_dev_f32(t.detach(), "parameter") for t in (qw, qb, kw, kb, vw, vb, m.proj_q.weight, m.proj_q.bias,
                                                      m.proj_kv.weight, m.proj_kv.bias, m.proj.weight, m.proj.bias)]
    keep.extend(ts)
    return _lib.AxvsTrajParams(*[t.data_ptr() for t in ts])


# ---------------------------------------------------------------------------------------------
# TrajectoryAttention   (WC/temporal_attention.py:20-76, TL:652-708)
# ---------------------------------------------------------------------------------------------
class TrajectoryAttention(nn.Module):
    def __init__(self, dim, num_heads=8, attn_drop=0., mfma_dtype: Optional[str] = None):
        super().__init__()
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = self.head_dim ** -0.5

        self.q = nn.Linear(dim, dim, bias=True)
        self.k = nn.Linear(dim, dim, bias=True)
        self.v = nn.Linear(dim, dim, bias=True)
        self.proj_q = nn.Linear(dim, dim, bias=True)
        self.proj_kv = nn.Linear(dim, dim * 2, bias=True)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)

        self.mfma_dtype = mfma_dtype
        self.return_attn = False     # the [(S h), N, T, L] map is opt-in: materialising it defeats the fusion
        self._packed: Optional[Tensor] = None
        self._packed_key = None

    def _dtype(self) -> str:
        return self.mfma_dtype or default_operand_dtype()

    def _pack(self) -> Tensor:
        dt = self._dtype()
        key = _param_key(self, dt)
        if self._packed is None or key != self._packed_key:
            L = _lib.lib()
            C_, dev = self.proj.weight.shape[0], self.proj.weight.device
            keep: list = []
            ps = _traj_struct(self, keep)
            buf = torch.empty(L.axvs_traj_packed_bytes(C_, self.num_heads), dtype=torch.uint8, device=dev)
            _lib.check(L.axvs_traj_pack(C.byref(ps), buf.data_ptr(), C_, self.num_heads, _lib.DTYPES[dt], _stream(dev)),
                       "axvs_traj_pack")
            self._packed, self._packed_key = buf, key
        return self._packed

    @_guarded
    def forward(self, query, key, value, num_frames=2):
        """query/key/value: [S, num_frames*L, C] -> (x [S, N, C], space_attn [(S h), N, T, L] or None)."""
        _require_eval(self)
        q, k, v = _dev_f32(query, "query"), _dev_f32(key, "key"), _dev_f32(value, "value")
        S, N, C_ = q.shape
        T = int(num_frames)
        if N % T:
            raise RuntimeError(f"tokens per sequence ({N}) must be a multiple of num_frames ({T})")
        Lx = N // T
        L = _lib.lib()
        out = torch.empty_like(q)
        attn = torch.empty(S * self.num_heads, N, T, Lx, dtype=torch.float32, device=q.device) if self.return_attn else None
        packed = self._pack()
        nws = L.axvs_traj_attn_workspace_bytes(S, T, Lx, C_, self.num_heads)
        ws = _workspace(q.device, nws)
        _lib.check(L.axvs_traj_attn_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), _ptr(attn),
                                        packed.data_ptr(), S, T, Lx, C_, self.num_heads, _lib.DTYPES[self._dtype()],
                                        ws.data_ptr(), ws.numel(), _stream(q.device)), "axvs_traj_attn_fwd")
        return out, attn


class _ffn_activation:
    """Context: tells the library which activation the layer's FFN uses (thread-local option; ReLU is the default and the only
    one the fused FFN kernels implement -- GELU takes the unfused LayerNorm / GEMM / GEMM / LayerNorm path)."""

    def __init__(self, name: str):
        if name == "glu":
            raise NotImplementedError("axial_vs_amd: activation='glu' halves the hidden width, which the reference's own linear2 "
                                      "(d_ffn inputs) cannot consume either (WC/temporal_attention.py:126,182)")
        self.gelu = name == "gelu"

    def __enter__(self):
        if self.gelu:
            _lib.lib().axvs_set_option(b"ffn_gelu", 1)

    def __exit__(self, *exc):
        if self.gelu:
            _lib.lib().axvs_set_option(b"ffn_gelu", 0)
        return False


def _get_activation_name(activation):
    if activation in ("relu", "gelu", "glu"):
        return activation
    raise RuntimeError(f"activation should be relu/gelu, not {activation}.")   # WC/temporal_attention.py:9-17


# ---------------------------------------------------------------------------------------------
# TemporalAxialTrajectoryAttentionLayer   (WC/temporal_attention.py:158-220, TL:730-791)
# ---------------------------------------------------------------------------------------------
class TemporalAxialTrajectoryAttentionLayer(nn.Module):
    def __init__(self, d_model=256, d_ffn=1024, dropout=0.0, attn_drop=0.0, activation="relu", n_heads=8,
                 mfma_dtype: Optional[str] = None):
        super().__init__()
        # self attention (note the reference's swapped naming: `dropout` feeds the attention-prob dropout)
        self.height_attn = TrajectoryAttention(d_model, n_heads, dropout, mfma_dtype)
        self.width_attn = TrajectoryAttention(d_model, n_heads, dropout, mfma_dtype)
        self.dropout1 = nn.Dropout(attn_drop)
        self.norm1 = nn.LayerNorm(d_model)
        # ffn
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.activation = _get_activation_name(activation)
        self.dropout2 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.dropout3 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)

        self.n_heads = n_heads
        self.mfma_dtype = mfma_dtype
        self.return_attn = False
        self.use_generated_pos = True   # a `pos` made by PositionEmbeddingSine3D is evaluated in-kernel instead of read (SineTag)
        # eval mode: None = fp32 output rows (the reference's type); torch.float16 / torch.bfloat16 = the layer's last kernel writes the
        # map in 16 bits (what a batch-sharded caller sends over the links: axial_vs_amd.dist, BASELINE config 5)
        self.out_dtype: Optional[torch.dtype] = None
        # train() mode: False (default) keeps the activations between forward and backward like the reference under autograd does
        # (44 C floats per token and layer: 0.74 GB at the metric shape -- sized for 288 GB of HBM); True: backward rebuilds them
        # from (src, pos, seed) first (+1 forward, nothing kept)
        self.recompute = False
        self.dropout_seed: Optional[int] = None   # train() mode: fixed dropout seed (tests); None = drawn from torch's CPU generator
        self._packed: Optional[Tensor] = None
        self._packed_key = None

    def _dtype(self) -> str:
        if self.linear1.in_features // self.n_heads > 32:
            return "f32"          # head_dim 64 (dim 256 / 4 heads, 512 / 8): built on the fp32 tier only; the 16-bit tiers stop at 32
        return self.mfma_dtype or _DEFAULT_DTYPE

    def _pack(self) -> Tensor:
        dt = self._dtype()
        if dt == "f32":
            raise NotImplementedError("axial_vs_amd: the fp32 tier (mfma_dtype='f32', head_dim 64) has no packed weights and no "
                                      "per-pass entry point (forward_pass / dist.offaxis_forward run on the 16-bit tier)")
        key = _param_key(self, dt)
        if self._packed is None or key != self._packed_key:
            if abs(self.norm1.eps - 1e-5) > 0 or abs(self.norm2.eps - 1e-5) > 0:
                raise NotImplementedError("axial_vs_amd: LayerNorm eps must be 1e-5")
            L = _lib.lib()
            C_, F, dev = self.linear1.in_features, self.linear1.out_features, self.linear1.weight.device
            keep: list = []
            ps = _lib.AxvsAxialLayerParams()
            ps.height_attn = _traj_struct(self.height_attn, keep)
            ps.width_attn = _traj_struct(self.width_attn, keep)
            for name, t in (("norm1_w", self.norm1.weight), ("norm1_b", self.norm1.bias),
                            ("linear1_w", self.linear1.weight), ("linear1_b", self.linear1.bias),
                            ("linear2_w", self.linear2.weight), ("linear2_b", self.linear2.bias),
                            ("norm2_w", self.norm2.weight), ("norm2_b", self.norm2.bias)):
                tt = _dev_f32(t.detach(), name)
                keep.append(tt)
                setattr(ps, name, tt.data_ptr())
            buf = torch.empty(L.axvs_axial_layer_packed_bytes(C_, self.n_heads, F), dtype=torch.uint8, device=dev)
            _lib.check(L.axvs_axial_layer_pack(C.byref(ps), buf.data_ptr(), C_, self.n_heads, F, _lib.DTYPES[dt],
                                               _stream(dev)), "axvs_axial_layer_pack")
            self._packed, self._packed_key = buf, key
        return self._packed

    @_guarded
    def forward(self, src: Tensor, pos: Tensor):
        """
        :param src: tensor of shape [B*T, H*W, C]
        :param pos: tensor of shape [B, T, H, W, C]
        :return: (src', height_traj_attn, width_traj_attn); the maps are None unless ``return_attn`` is set

        ``train()`` mode: the differentiable training tier (dropout active, fp32, `axial_vs_amd.training`); ``eval()`` mode: the
        fused 16-bit MFMA inference tier (no autograd graph).
        """
        if self.training:
            if self.return_attn:
                raise NotImplementedError("axial_vs_amd: attention maps are an eval-mode output (visualize_attn)")
            from .training import axial_layer_train
            return axial_layer_train(self, src, pos, recompute=self.recompute), None, None
        if self._dtype() == "f32":
            # fp32 tier: the training tier's forward with dropout off -- fp32 MFMA (v_mfma_f32_16x16x4_f32) attention, fp32 GEMMs, fp32
            # everywhere (1e-6 against the float64 oracle).  ~12x the time of the 16-bit tier: for operands beyond the fp16 range
            # (enable_range_check reports them) or callers that need more than the 1e-3 bar.  head_dim in {8, 16, 32, 64}, T <= 16
            # (head_dim 64 always runs here).
            if self.return_attn:
                raise NotImplementedError("axial_vs_amd: attention maps are an output of the 16-bit tier (mfma_dtype 'f16' / 'bf16')")
            if self.out_dtype is not None:
                raise NotImplementedError("axial_vs_amd: out_dtype (a 16-bit output map written by the kernel epilogue) exists on the 16-bit tier only; "
                                          "the fp32 tier (mfma_dtype='f32') returns fp32 rows")
            from .training import axial_layer_train
            with torch.no_grad(), torch.autocast(device_type="cuda", enabled=False):      # (the fp32 tier stays fp32 under autocast)
                return axial_layer_train(self, src, pos, dropout=False), None, None
        B, T, H, W = pos.shape[:4]
        s, p = _dev_f32(src, "src"), _dev_f32(pos, "pos")
        C_ = s.shape[-1]
        if s.numel() != B * T * H * W * C_ or p.shape[-1] != C_:
            raise RuntimeError(f"src {tuple(src.shape)} does not match pos {tuple(pos.shape)}")
        L = _lib.lib()
        F = self.linear1.out_features
        if self.out_dtype is not None:
            # the output map in 16 bits, written by the epilogue of the kernel that ends the layer (library option "layer_out_dtype")
            code = {torch.float16: 1, torch.bfloat16: 2}.get(self.out_dtype)
            if code is None:
                raise ValueError("out_dtype: None (fp32, the reference's type), torch.float16 or torch.bfloat16")
            if self.return_attn:
                raise NotImplementedError("axial_vs_amd: out_dtype and return_attn exclude each other")
            out = torch.empty(s.shape, dtype=self.out_dtype, device=s.device)
            _lib.check(L.axvs_set_option(b"layer_out_dtype", code), "axvs_set_option")
            try:
                return self._launch(L, s, p, pos, out, None, None, B, T, H, W, C_, F)
            finally:
                L.axvs_set_option(b"layer_out_dtype", 0)
        out = torch.empty_like(s)
        ha = wa = None
        if self.return_attn:
            ha = torch.empty(B * W * self.n_heads, T * H, T, H, dtype=torch.float32, device=s.device)
            wa = torch.empty(B * H * self.n_heads, T * W, T, W, dtype=torch.float32, device=s.device)
        return self._launch(L, s, p, pos, out, ha, wa, B, T, H, W, C_, F)

    def _launch(self, L, s, p, pos, out, ha, wa, B, T, H, W, C_, F):
        packed = self._pack()
        sh = _stream(s.device)                     # the current stream's handle, looked up once per call
        _select_sync_words(s.device, sh)
        tag = _sine_tag(pos) if self.use_generated_pos else None
        if tag is not None:
            sp = _lib.AxvsSinePos3D(tag.temperature, int(tag.normalize), tag.scale, tag.level.data_ptr() if tag.level is not None else None)
            ws = _workspace(s.device, L.axvs_axial_layer_workspace_bytes_ex(B, T, H, W, C_, self.n_heads, F, int(self.return_attn), 1), sh)
            _lib.check(L.axvs_axial_layer_fwd_sine3d(s.data_ptr(), C.byref(sp), out.data_ptr(), packed.data_ptr(), B, T, H, W, C_,
                                                     self.n_heads, F, _lib.DTYPES[self._dtype()], ws.data_ptr(), ws.numel(),
                                                     _ptr(ha), _ptr(wa), sh), "axvs_axial_layer_fwd_sine3d")
            return out, ha, wa
        nws = L.axvs_axial_layer_workspace_bytes_ex(B, T, H, W, C_, self.n_heads, F, int(self.return_attn), 0)
        ws = _workspace(s.device, nws, sh)
        _lib.check(L.axvs_axial_layer_fwd(s.data_ptr(), p.data_ptr(), out.data_ptr(), packed.data_ptr(), B, T, H, W, C_,
                                          self.n_heads, F, _lib.DTYPES[self._dtype()], ws.data_ptr(), ws.numel(),
                                          _ptr(ha), _ptr(wa), sh), "axvs_axial_layer_fwd")
        return out, ha, wa


    # in-place levels: the row-addressed temporaries of the layer take the FRAME STRIDE of the token buffer (sparse buffers of
    # (B*T - 1) * stride + H*W rows).  A small level inside a long buffer would blow the workspace up (B*T = 32 frames of 21504
    # tokens: 680 MB per temporary for a 32 x 32 level against 32 MB): beyond this overhead the split / cat data flow runs.
    _IN_PLACE_WS_OVERHEAD = 256 << 20

    def can_run_in_place(self, pos: Tensor, frame_stride_rows: Optional[int] = None) -> bool:
        """Whether `forward_level_in_place` applies: eval mode, 16-bit fused tier, generated positions, no attention maps -- and, when
        the caller names the frame stride of its token buffer, row indices within 32 bits and a workspace not more than
        `_IN_PLACE_WS_OVERHEAD` bytes above the contiguous call's."""
        if self.training or self.return_attn or not self.use_generated_pos or self._dtype() == "f32" or _sine_tag(pos) is None:
            return False
        if _has_hooks(self):              # forward hooks see forward()'s per-level tensors: the split / cat data flow runs for them
            return False
        L = _lib.lib()
        C_, F = self.linear1.in_features, self.linear1.out_features
        if not L.axvs_axial_layer_strided_ok(C_, self.n_heads, F):
            return False
        if frame_stride_rows is not None:
            B, T, H, W = pos.shape[:4]
            span = (B * T - 1) * int(frame_stride_rows) + H * W
            if span > (2 ** 31 - 1) // 64:
                return False
            strided = L.axvs_axial_layer_workspace_bytes_strided(B, T, H, W, C_, self.n_heads, F, int(frame_stride_rows))
            natural = L.axvs_axial_layer_workspace_bytes_ex(B, T, H, W, C_, self.n_heads, F, 0, 1)
            if strided > natural + self._IN_PLACE_WS_OVERHEAD:
                return False
        return True

    @_guarded
    def forward_level_in_place(self, tokens: Tensor, start: int, pos: Tensor) -> None:
        """The layer applied IN PLACE to rows [start, start + H*W) of every frame of `tokens` [B*T, S, C] -- one level of the pixel
        decoder's concatenated token buffer, where the reference splits the level out, runs the layer and concatenates again
        (WC/msdeformattn.py:258-264): no copy out, no copy back.  `pos` [B, T, H, W, C] made by PositionEmbeddingSine3D."""
        B, T, H, W = pos.shape[:4]
        tag = _sine_tag(pos)
        if tokens.dtype != torch.float32 or not tokens.is_cuda or not tokens.is_contiguous() or tokens.dim() != 3:
            raise RuntimeError("tokens: contiguous fp32 [B*T, S, C] on the GPU")
        BT, S, C_ = tokens.shape
        if BT != B * T or start < 0 or start + H * W > S or pos.shape[-1] != C_ or tag is None:
            raise RuntimeError(f"tokens {tuple(tokens.shape)} / start {start} do not match pos {tuple(pos.shape)}")
        L = _lib.lib()
        F = self.linear1.out_features
        packed = self._pack()
        sh = _stream(tokens.device)
        _select_sync_words(tokens.device, sh)
        sp = _lib.AxvsSinePos3D(tag.temperature, int(tag.normalize), tag.scale, tag.level.data_ptr() if tag.level is not None else None)
        ws = _workspace(tokens.device, L.axvs_axial_layer_workspace_bytes_strided(B, T, H, W, C_, self.n_heads, F, S), sh)
        base = tokens.data_ptr() + start * C_ * 4
        _lib.check(L.axvs_axial_layer_fwd_sine3d_strided(base, C.byref(sp), base, packed.data_ptr(), B, T, H, W, C_, self.n_heads, F,
                                                         _lib.DTYPES[self._dtype()], S, ws.data_ptr(), ws.numel(), sh),
                   "axvs_axial_layer_fwd_sine3d_strided")

    @_guarded
    def forward_pass(self, x: Tensor, pos: Tensor, which: int) -> Tensor:
        """One axial pass on a LOCAL block [B,T,H,W,C] of the token grid (`pos`: the matching block of the embedding):
        which = 0: x + height_attn(x + pos, x) -- valid on any block of columns;
        which = 1: norm2(FFN(norm1(x + width_attn(x + pos, x)))) -- valid on any block of rows.
        The building block of `axial_vs_amd.dist.offaxis_forward` (one clip over several GPUs)."""
        _require_eval(self)
        xs, ps = _dev_f32(x, "x"), _dev_f32(pos, "pos")
        if xs.dim() != 5 or xs.shape != ps.shape:
            raise RuntimeError(f"x {tuple(x.shape)} and pos {tuple(pos.shape)} must be equal [B,T,H,W,C] blocks")
        B, T, H, W, C_ = xs.shape
        L = _lib.lib()
        F = self.linear1.out_features
        out = torch.empty_like(xs)
        packed = self._pack()
        _select_sync_words(xs.device)
        ws = _workspace(xs.device, L.axvs_axial_layer_workspace_bytes_ex(B, T, H, W, C_, self.n_heads, F, 0, 0))
        _lib.check(L.axvs_axial_pass_fwd(xs.data_ptr(), ps.data_ptr(), out.data_ptr(), packed.data_ptr(), int(which), B, T, H, W, C_,
                                         self.n_heads, F, _lib.DTYPES[self._dtype()], ws.data_ptr(), ws.numel(), _stream(xs.device)),
                   "axvs_axial_pass_fwd")
        return out


class TemporalTrajectoryAttentionLayer(nn.Module):
    """Full T*H*W trajectory attention (WC/temporal_attention.py:103-155, `temporal_attn_type="trajectory"`): ONE
    TrajectoryAttention over all tokens of a clip, then norm1 -> FFN -> norm2.  Unused by every shipped config (all select
    'axial-trajectory'); same parameter names; returns (src', None, None) like the reference."""

    def __init__(self, d_model=256, d_ffn=1024, dropout=0.0, attn_drop=0.0, activation="relu", n_heads=8,
                 mfma_dtype: Optional[str] = None):
        super().__init__()
        self.temporal_attn = TrajectoryAttention(d_model, n_heads, dropout, mfma_dtype)
        self.dropout1 = nn.Dropout(attn_drop)
        self.norm1 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.activation = _get_activation_name(activation)
        self.dropout2 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.dropout3 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)
        self.n_heads = n_heads
        self.mfma_dtype = mfma_dtype
        self._packed: Optional[Tensor] = None
        self._packed_key = None

    def _dtype(self) -> str:
        return self.mfma_dtype or default_operand_dtype()

    def _pack(self) -> Tensor:
        dt = self._dtype()
        key = _param_key(self, dt)
        if self._packed is None or key != self._packed_key:
            L = _lib.lib()
            C_, F, dev = self.linear1.in_features, self.linear1.out_features, self.linear1.weight.device
            keep: list = []
            ps = _lib.AxvsTrajLayerParams()
            ps.temporal_attn = _traj_struct(self.temporal_attn, keep)
            for name, t in (("norm1_w", self.norm1.weight), ("norm1_b", self.norm1.bias),
                            ("linear1_w", self.linear1.weight), ("linear1_b", self.linear1.bias),
                            ("linear2_w", self.linear2.weight), ("linear2_b", self.linear2.bias),
                            ("norm2_w", self.norm2.weight), ("norm2_b", self.norm2.bias)):
                tt = _dev_f32(t.detach(), name)
                keep.append(tt)
                setattr(ps, name, tt.data_ptr())
            buf = torch.empty(L.axvs_traj_layer_packed_bytes(C_, self.n_heads, F), dtype=torch.uint8, device=dev)
            _lib.check(L.axvs_traj_layer_pack(C.byref(ps), buf.data_ptr(), C_, self.n_heads, F, _lib.DTYPES[dt], _stream(dev)),
                       "axvs_traj_layer_pack")
            torch.cuda.current_stream(dev).synchronize()
            self._packed, self._packed_key = buf, key
        return self._packed

    @_guarded
    def forward(self, src: Tensor, pos: Tensor):
        """
        :param src: tensor of shape [B*T, H*W, C]
        :param pos: tensor of shape [B, T, H, W, C]
        """
        _require_eval(self)
        B, T, H, W = pos.shape[:4]
        s, p = _dev_f32(src, "src"), _dev_f32(pos, "pos")
        C_ = s.shape[-1]
        if s.numel() != B * T * H * W * C_ or p.shape[-1] != C_:
            raise RuntimeError(f"src {tuple(src.shape)} does not match pos {tuple(pos.shape)}")
        L = _lib.lib()
        F = self.linear1.out_features
        out = torch.empty_like(s)
        packed = self._pack()
        ws = _workspace(s.device, L.axvs_traj_layer_workspace_bytes(B, T, H * W, C_, self.n_heads, F))
        _lib.check(L.axvs_traj_layer_fwd(s.data_ptr(), p.data_ptr(), out.data_ptr(), packed.data_ptr(), B, T, H * W, C_, self.n_heads, F,
                                         _lib.DTYPES[self._dtype()], ws.data_ptr(), ws.numel(), _stream(s.device)), "axvs_traj_layer_fwd")
        return out, None, None


# ---------------------------------------------------------------------------------------------
# TemporalEncoder   (WC/temporal_attention.py:79-100;  TL:711-727)
# ---------------------------------------------------------------------------------------------
class TemporalEncoder(nn.Module):
    def __init__(self, d_model=256, d_ffn=1024, dropout=0.0, attn_drop=0.0, activation="relu", n_heads=8,
                 temporal_attn_type="trajectory", num_temporal_layer=2, mfma_dtype: Optional[str] = None):
        super().__init__()
        if temporal_attn_type == "trajectory":
            self.temporal_layers = nn.ModuleList([TemporalTrajectoryAttentionLayer(
                d_model, d_ffn, dropout, attn_drop, activation, n_heads, mfma_dtype) for _ in range(num_temporal_layer)])
        elif temporal_attn_type == "axial-trajectory":
            self.temporal_layers = nn.ModuleList([TemporalAxialTrajectoryAttentionLayer(
                d_model, d_ffn, dropout, attn_drop, activation, n_heads, mfma_dtype) for _ in range(num_temporal_layer)])
        # any other string: no layers are created, exactly like the reference (:85-88)

    def prepack(self) -> None:
        """Pack (or re-pack after a weight change) every layer's weights NOW, on the current stream.  A caller that is about to run
        this encoder on several streams at once calls it first: the pack kernels and the packed buffer's allocation then belong to
        the stream every fork waits on, instead of to whichever stream reaches `_pack()` first."""
        if self.training:
            return
        for layer in getattr(self, "temporal_layers", ()):
            if layer._dtype() != "f32":
                layer._pack()

    def forward(self, src: Tensor, pos: Tensor):
        """
        :param src: tensor of shape [B*T, H*W, C]
        :param pos: tensor of shape [B, T, H, W, C]
        """
        for layer in self.temporal_layers:
            src, height_traj_attn, width_traj_attn = layer(src, pos)
        return src, height_traj_attn, width_traj_attn

    def can_run_in_place(self, pos: Tensor, frame_stride_rows: Optional[int] = None) -> bool:
        layers = getattr(self, "temporal_layers", None)
        return (not self.training and layers is not None and len(layers) > 0 and not _has_hooks(self)
                and all(isinstance(l, TemporalAxialTrajectoryAttentionLayer) and l.can_run_in_place(pos, frame_stride_rows) for l in layers))

    def forward_level_in_place(self, tokens: Tensor, start: int, pos: Tensor) -> None:
        """Every layer of the encoder applied in place to one level of the concatenated token buffer (see the layer's method)."""
        for layer in self.temporal_layers:
            layer.forward_level_in_place(tokens, start, pos)


class TubeLinkTemporalEncoder(nn.Module):
    """Tube-Link flavour (TL:711-727): axial layers only, returns ``src`` alone."""

    def __init__(self, d_model=256, d_ffn=1024, dropout=0.0, attn_drop=0.0, activation="relu", n_heads=8,
                 num_temporal_layer=2, mfma_dtype: Optional[str] = None):
        super().__init__()
        self.temporal_layers = nn.ModuleList([TemporalAxialTrajectoryAttentionLayer(
            d_model, d_ffn, dropout, attn_drop, activation, n_heads, mfma_dtype) for _ in range(num_temporal_layer)])

    def forward(self, src: Tensor, pos: Tensor):
        for layer in self.temporal_layers:
            src = layer(src, pos)[0]
        return src


# ---------------------------------------------------------------------------------------------
# PositionEmbeddingSine3D   (WC/pos_embeddings.py:68-130)
# ---------------------------------------------------------------------------------------------
class PositionEmbeddingSine3D(nn.Module):
    def __init__(self, num_pos_feats=64, temperature=10000, normalize=False, scale=None):
        super().__init__()
        self.num_pos_feats = num_pos_feats
        self.temperature = temperature
        self.normalize = normalize
        if scale is not None and normalize is False:
            raise ValueError("normalize should be True if scale is passed")
        if scale is None:
            scale = 2 * math.pi
        self.scale = scale
        self.is_3d = True

    @torch.no_grad()
    def channels_last(self, B: int, T: int, H: int, W: int, device) -> Tensor:
        """[B,T,H,W,C] fp32 -- the layout the trajectory layers consume."""
        Cc = 2 * self.num_pos_feats
        pos = torch.empty(B, T, H, W, Cc, dtype=torch.float32, device=device)
        with _on(pos.device):
            _lib.check(_lib.lib().axvs_pos3d(pos.data_ptr(), B, T, H, W, Cc, float(self.temperature), int(self.normalize),
                                             float(self.scale), _stream(pos.device)), "axvs_pos3d")
        return tag_sine3d(pos, self.temperature, self.normalize, self.scale)

    @torch.no_grad()
    def channels_last_with_level(self, B: int, T: int, H: int, W: int, level: Tensor) -> Tensor:
        """pos + level[None,None,None,None,:] (the decoder's level_embed_3d, WC/msdeformattn.py:112-115), tagged so that the
        trajectory layers can regenerate it in-kernel."""
        lv = level.detach().to(torch.float32).contiguous()
        pos = self.channels_last(B, T, H, W, lv.device)
        with _on(pos.device):
            _lib.check(_lib.lib().axvs_add_channel_vector(pos.data_ptr(), lv.data_ptr(), pos.numel(), pos.shape[-1], _stream(pos.device)),
                       "axvs_add_channel_vector")
        return tag_sine3d(pos, self.temperature, self.normalize, self.scale, lv)

    @torch.no_grad()
    def forward(self, x, mask=None, fmt="btchw"):
        assert x.dim() == 5, f"{x.shape} should be a 5-dimensional Tensor, got {x.dim()}-dimensional Tensor instead"
        if fmt == "btchw":
            B, T, _, H, W = x.shape
        elif fmt == "bcthw":
            B, _, T, H, W = x.shape
        else:
            raise ValueError(f"Invalid format given: {fmt})")
        if not x.is_cuda:
            raise RuntimeError("axial_vs_amd: x must be a GPU tensor; there is no CPU fallback")
        if mask is not None:          # padding mask [B,T,H,W], True = padded (:96-106); a plain tensor (no SineTag: not regenerable)
            if tuple(mask.shape) != (B, T, H, W):
                raise RuntimeError(f"mask {tuple(mask.shape)} must be [B,T,H,W] = {(B, T, H, W)}")
            m8 = mask.to(device=x.device, dtype=torch.uint8).contiguous()
            Cc = 2 * self.num_pos_feats
            pos = torch.empty(B, T, H, W, Cc, dtype=torch.float32, device=x.device)
            with _on(pos.device):
                _lib.check(_lib.lib().axvs_pos3d_masked(pos.data_ptr(), m8.data_ptr(), B, T, H, W, Cc, float(self.temperature),
                                                        int(self.normalize), float(self.scale), _stream(pos.device)), "axvs_pos3d_masked")
        else:
            pos = self.channels_last(B, T, H, W, x.device)
        return pos.permute(0, 1, 4, 2, 3) if fmt == "btchw" else pos.permute(0, 4, 1, 2, 3)


# ---------------------------------------------------------------------------------------------
# [B,T,C,H,W] convenience surface named by the north star
# ---------------------------------------------------------------------------------------------
class AxialTrajectoryAttention5D(nn.Module):
    """forward(x: [B,T,C,H,W]) -> [B,T,C,H,W]: builds the 3-D sine embedding (+ optional learned level embedding,
    WC/msdeformattn.py:112-115) and runs a TemporalEncoder of axial-trajectory layers."""

    def __init__(self, d_model=256, d_ffn=1024, n_heads=8, num_temporal_layer=1, level_embed: bool = False,
                 mfma_dtype: Optional[str] = None):
        super().__init__()
        self.encoder = TemporalEncoder(d_model, d_ffn, 0.0, 0.0, "relu", n_heads, "axial-trajectory", num_temporal_layer,
                                       mfma_dtype)
        self.pos_embed = PositionEmbeddingSine3D(d_model // 2, normalize=True)
        self.level_embed_3d = nn.Parameter(torch.zeros(d_model)) if level_embed else None
        self._pos_cache: Dict[Tuple, Tensor] = {}

    def position(self, B, T, H, W, device) -> Tensor:
        lv = self.level_embed_3d
        key = (B, T, H, W, str(device), None if lv is None else (lv.data_ptr(), lv._version))
        pos = self._pos_cache.get(key)
        if pos is None:
            pos = (self.pos_embed.channels_last(B, T, H, W, device) if lv is None
                   else self.pos_embed.channels_last_with_level(B, T, H, W, lv))
            self._pos_cache = {key: pos}
        return pos

    def forward(self, x: Tensor) -> Tensor:
        B, T, C_, H, W = x.shape
        src = x.permute(0, 1, 3, 4, 2).reshape(B * T, H * W, C_)
        out, _, _ = self.encoder(src, self.position(B, T, H, W, x.device))
        return out.reshape(B, T, H, W, C_).permute(0, 1, 4, 2, 3)


class GraphedForward:
    """Replays `module(*inputs)` from a captured HIP graph: one hipGraphLaunch per step instead of a Python round trip per
    kernel (the library's launches are asynchronous on the capture stream and never allocate or synchronise, so the whole
    forward is capturable; weights must be packed and workspaces allocated first, which the warm-up call does).

        g = GraphedForward(layer, src, pos)      # captures layer(src, pos) on static copies of the inputs
        out = g(src2, pos2)                      # copies the new inputs into the static buffers, replays, returns g.out
        out = g()                                # replay on the current contents (e.g. inputs produced in place)
    """

    def __init__(self, module: nn.Module, *inputs: Tensor, warmup: int = 2):
        if not all(t.is_cuda for t in inputs):
            raise RuntimeError("GraphedForward: inputs must be CUDA tensors")
        self.module = module
        self.inputs = tuple(t.detach().clone() for t in inputs)
        dev = self.inputs[0].device
        # The graph owns its arrival counters (merged q/k/v + trajectory launches): allocated and zeroed NOW, eagerly -- an allocation
        # during the capture would come from the graph's private pool with its zero-fill recorded as a graph node instead of executed,
        # and every graph captured on torch's shared capture stream would bake the same pointer in (graphs replayed concurrently on
        # different streams must not share counters: include/axvs.h, "one buffer serves one stream at a time").
        self._sync = [torch.zeros(_SYNC_WORDS, dtype=torch.int32, device=dev) for _ in range(4)]
        torch.cuda.synchronize(dev)
        prev_override = getattr(_sync_tls, "override", None)
        # (warm-up and capture run on different streams: each gets its own set; sets left in the pool are simply kept)
        self._ws: Dict[Tuple[str, int], Tensor] = {}        # ... and its scratch buffers (replays of different graphs may overlap in time)
        _sync_tls.override = {"pool": list(self._sync), "map": {}, "ws": self._ws}
        try:
            side = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):                 # warm-up off the default stream, as graph capture requires
                for _ in range(max(warmup, 1)):
                    module(*self.inputs)
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.out = module(*self.inputs)
        finally:
            _sync_tls.override = prev_override
            _sync_tls.key = None                          # the next eager call registers its own (device, stream) words again
        _live_graphs.add(self)

    def _timed_out(self) -> bool:
        w = _status_words.get(self.inputs[0].device.index)
        return w is not None and bool(int(w[0]) & 4)

    def __call__(self, *inputs: Tensor):
        # A replay bypasses the library's entry gate: read the (host-resident) status word here.  A hand-off timeout of an earlier replay left
        # this graph's arrival counters non-zero -- every further replay would time out again: put them in order and fail loudly (no sync).
        if self._timed_out():
            _on_sync_timeout()
            raise RuntimeError("axial_vs_amd: a merged q/k/v + trajectory launch of an earlier graph replay gave up waiting for its sibling row "
                               "tiles (AXVS_STATUS_SYNC_TIMEOUT): that replay's outputs are NaN rows.  The arrival counters have been zeroed; replay again")
        for dst, src in zip(self.inputs, inputs):
            dst.copy_(src, non_blocking=True)
        self.graph.replay()
        if _handoff_policy[0] == "verify":
            dev = self.inputs[0].device
            torch.cuda.current_stream(dev).synchronize()
            if self._timed_out():                         # recover transparently: the same forward, eagerly, two launches per pass
                _on_sync_timeout()
                _handoff_recoveries[0] += 1
                L = _lib.lib()
                _lib.check(L.axvs_set_option(b"no_merge_qkv", 1), "axvs_set_option")
                try:
                    res = self.module(*self.inputs)
                    for dst, src in zip(self.out if isinstance(self.out, (tuple, list)) else (self.out,),
                                        res if isinstance(res, (tuple, list)) else (res,)):
                        if isinstance(dst, Tensor) and isinstance(src, Tensor):
                            dst.copy_(src)
                    torch.cuda.current_stream(dev).synchronize()
                finally:
                    L.axvs_set_option(b"no_merge_qkv", 0)
        return self.out
