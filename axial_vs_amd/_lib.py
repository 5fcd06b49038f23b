"""ctypes binding of libaxvs.so (C-ABI declared in include/axvs.h).

There is no CPU or PyTorch fallback: if the HIP library is missing, importing succeeds but the
first call raises, loudly.  Build it with ``python -c "import __graft_entry__ as g; g.build()"``.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# AXVS_LIB_PATH: diagnostic builds of the same library (tools/ only: -DAXVS_STAMPS, ablations)
LIB_PATH = os.environ.get("AXVS_LIB_PATH") or os.path.join(_HERE, "libaxvs.so")

AXVS_F16 = 0
AXVS_BF16 = 1
DTYPES = {"f16": AXVS_F16, "fp16": AXVS_F16, "float16": AXVS_F16, "bf16": AXVS_BF16, "bfloat16": AXVS_BF16}

_fp = C.c_void_p  # device pointers travel as integers


class AxvsTrajParams(C.Structure):
    _fields_ = [(n, _fp) for n in ("q_w", "q_b", "k_w", "k_b", "v_w", "v_b", "proj_q_w", "proj_q_b",
                                   "proj_kv_w", "proj_kv_b", "proj_w", "proj_b")]


class AxvsAxialLayerParams(C.Structure):
    _fields_ = [("height_attn", AxvsTrajParams), ("width_attn", AxvsTrajParams)] + \
               [(n, _fp) for n in ("norm1_w", "norm1_b", "linear1_w", "linear1_b", "linear2_w", "linear2_b",
                                   "norm2_w", "norm2_b")]


class AxvsTrajLayerParams(C.Structure):
    _fields_ = [("temporal_attn", AxvsTrajParams)] + [(n, _fp) for n in ("norm1_w", "norm1_b", "linear1_w", "linear1_b", "linear2_w",
                                                                          "linear2_b", "norm2_w", "norm2_b")]


class AxvsSinePos3D(C.Structure):
    _fields_ = [("temperature", C.c_float), ("normalize", C.c_int), ("scale", C.c_float), ("level_embed", _fp)]


class AxvsBN(C.Structure):
    _fields_ = [(n, _fp) for n in ("w", "b", "mean", "var")]


class AxvsCCLayerParams(C.Structure):
    _fields_ = [("attn", AxvsTrajParams), ("norm_w", _fp), ("norm_b", _fp), ("aspp_w", _fp * 3), ("aspp_b", _fp * 3),
                ("aspp_proj_w", _fp), ("aspp_norm_w", _fp), ("aspp_norm_b", _fp), ("conv_norm_w", _fp), ("conv_norm_b", _fp)]


class AxvsCCHeadParams(C.Structure):
    _fields_ = [("class_proj_w", _fp), ("class_proj_bn", AxvsBN), ("mask_proj_w", _fp), ("mask_proj_bn", AxvsBN),
                ("mask_head_w", _fp), ("mask_head_bn", AxvsBN), ("class_head_w", _fp), ("class_head_b", _fp),
                ("act_head_w", _fp), ("act_head_b", _fp), ("pixel_bn", AxvsBN)]


class AxvsBNGrads(C.Structure):
    _fields_ = [("w", _fp), ("b", _fp)]


class AxvsCCHeadGrads(C.Structure):
    _fields_ = [("class_proj_w", _fp), ("class_proj_bn", AxvsBNGrads), ("mask_proj_w", _fp), ("mask_proj_bn", AxvsBNGrads),
                ("mask_head_w", _fp), ("mask_head_bn", AxvsBNGrads), ("class_head_w", _fp), ("class_head_b", _fp),
                ("act_head_w", _fp), ("act_head_b", _fp), ("pixel_bn", AxvsBNGrads)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


class AxvsCCTrainCfg(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("B", "Q", "Tc", "V", "H", "W", "K1", "num_layers")] + \
               [("rates", C.c_int * 3), ("p_attn_drop", C.c_float), ("p_aspp_drop", C.c_float), ("seed", C.c_uint),
                ("allreduce", ALLREDUCE_FN), ("allreduce_user", C.c_void_p), ("chain_only", C.c_int)]


class AxvsTLHeadParams(C.Structure):
    _fields_ = [(n, _fp) for n in ("post_norm_w", "post_norm_b", "activation_proj_w", "activation_proj_b", "cls_embed_w",
                                   "cls_embed_b")] + [("mask_embed_w", _fp * 3), ("mask_embed_b", _fp * 3)]


class AxvsMsdaParams(C.Structure):
    _fields_ = [(n, _fp) for n in ("value_proj_w", "value_proj_b", "sampling_offsets_w", "sampling_offsets_b",
                                   "attention_weights_w", "attention_weights_b", "output_proj_w", "output_proj_b")]


class AxvsMsdaLayerParams(C.Structure):
    _fields_ = [("self_attn", AxvsMsdaParams)] + [(n, _fp) for n in ("norm1_w", "norm1_b", "linear1_w", "linear1_b", "linear2_w",
                                                                      "linear2_b", "norm2_w", "norm2_b")]


class AxvsConvGnParams(C.Structure):
    _fields_ = [(n, _fp) for n in ("conv_w", "conv_b", "gn_w", "gn_b")]


# name -> (restype, argtypes); must list every symbol of include/axvs.h
SIGNATURES = {
    "axvs_version": (C.c_int, []),
    "axvs_has_bf16": (C.c_int, []),
    "axvs_last_error": (C.c_char_p, []),
    "axvs_set_status_buffer": (C.c_int, [_fp]),
    "axvs_check_status": (C.c_int, []),
    "axvs_set_sync_buffer": (C.c_int, [_fp, C.c_size_t]),
    "axvs_profile_stages": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "axvs_profile_stage_count": (C.c_int, []),
    "axvs_profile_stage_name": (C.c_char_p, [C.c_int]),
    "axvs_set_option": (C.c_int, [C.c_char_p, C.c_int]),
    "axvs_traj_packed_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "axvs_traj_pack": (C.c_int, [C.POINTER(AxvsTrajParams), _fp, C.c_int, C.c_int, C.c_int, _fp]),
    "axvs_axial_layer_packed_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "axvs_axial_layer_pack": (C.c_int, [C.POINTER(AxvsAxialLayerParams), _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    "axvs_traj_attn_workspace_bytes": (C.c_size_t, [C.c_int] * 5),
    "axvs_traj_attn_fwd": (C.c_int, [_fp, _fp, _fp, _fp, _fp, _fp] + [C.c_int] * 6 + [_fp, C.c_size_t, _fp]),
    "axvs_axial_layer_workspace_bytes": (C.c_size_t, [C.c_int] * 7),
    "axvs_axial_layer_fwd": (C.c_int, [_fp, _fp, _fp, _fp] + [C.c_int] * 8 + [_fp, C.c_size_t, _fp, _fp, _fp]),
    "axvs_traj_layer_packed_bytes": (C.c_size_t, [C.c_int] * 3),
    "axvs_traj_layer_pack": (C.c_int, [C.POINTER(AxvsTrajLayerParams), _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    "axvs_traj_layer_workspace_bytes": (C.c_size_t, [C.c_int] * 6),
    "axvs_traj_layer_fwd": (C.c_int, [_fp, _fp, _fp, _fp] + [C.c_int] * 7 + [_fp, C.c_size_t, _fp]),
    "axvs_axial_layer_train_saved_bytes": (C.c_size_t, [C.c_int] * 7),
    "axvs_axial_layer_train_scratch_bytes": (C.c_size_t, [C.c_int] * 8),
    "axvs_axial_layer_train_fwd": (C.c_int, [_fp, _fp, _fp, C.POINTER(AxvsAxialLayerParams)] + [C.c_int] * 7 +
                                   [C.c_float, C.c_float, C.c_uint, _fp, C.c_size_t, _fp, C.c_size_t, _fp]),
    "axvs_axial_layer_train_bwd": (C.c_int, [_fp, _fp, _fp, C.POINTER(AxvsAxialLayerParams), C.POINTER(AxvsAxialLayerParams), _fp, _fp] +
                                   [C.c_int] * 7 + [C.c_float, C.c_float, C.c_uint, C.c_int, _fp, C.c_size_t, _fp, C.c_size_t, _fp]),
    "axvs_cc_module_train_saved_bytes": (C.c_size_t, [C.POINTER(AxvsCCTrainCfg)]),
    "axvs_cc_module_train_scratch_bytes": (C.c_size_t, [C.POINTER(AxvsCCTrainCfg), C.c_int]),
    "axvs_cc_module_train_bn_stats_floats": (C.c_size_t, [C.POINTER(AxvsCCTrainCfg)]),
    "axvs_cc_module_train_fwd": (C.c_int, [_fp] * 5 + [C.POINTER(AxvsCCLayerParams), C.POINTER(AxvsCCHeadParams), C.POINTER(AxvsCCTrainCfg),
                                           _fp, C.c_size_t, _fp, C.c_size_t, _fp]),
    "axvs_cc_module_train_bwd": (C.c_int, [_fp] * 4 + [C.POINTER(AxvsCCLayerParams), C.POINTER(AxvsCCHeadParams), C.POINTER(AxvsCCLayerParams),
                                           C.POINTER(AxvsCCHeadGrads), _fp, C.POINTER(AxvsCCTrainCfg), _fp, C.c_size_t, _fp, C.c_size_t, _fp]),
    "axvs_cc_layers_train_fwd": (C.c_int, [_fp, _fp, C.POINTER(AxvsCCLayerParams), C.POINTER(AxvsCCTrainCfg), _fp, C.c_size_t, _fp, C.c_size_t, _fp]),
    "axvs_cc_layers_train_bwd": (C.c_int, [_fp, _fp, C.POINTER(AxvsCCLayerParams), C.POINTER(AxvsCCLayerParams), _fp, C.POINTER(AxvsCCTrainCfg),
                                           _fp, C.c_size_t, _fp, C.c_size_t, _fp]),
    "axvs_axial_pass_fwd": (C.c_int, [_fp, _fp, _fp, _fp] + [C.c_int] * 9 + [_fp, C.c_size_t, _fp]),
    "axvs_axial_layer_workspace_bytes_ex": (C.c_size_t, [C.c_int] * 9),
    "axvs_axial_layer_sine3d_workspace_bytes": (C.c_size_t, [C.c_int] * 7),
    "axvs_axial_layer_fwd_sine3d": (C.c_int, [_fp, C.POINTER(AxvsSinePos3D), _fp, _fp] + [C.c_int] * 8 + [_fp, C.c_size_t, _fp, _fp, _fp]),
    "axvs_axial_layer_strided_ok": (C.c_int, [C.c_int] * 3),
    "axvs_axial_layer_workspace_bytes_strided": (C.c_size_t, [C.c_int] * 7 + [C.c_longlong]),
    "axvs_axial_layer_fwd_sine3d_strided": (C.c_int, [_fp, C.POINTER(AxvsSinePos3D), _fp, _fp] + [C.c_int] * 8 + [C.c_longlong, _fp, C.c_size_t, _fp]),
    "axvs_ffn_workspace_bytes": (C.c_size_t, [C.c_longlong, C.c_int, C.c_int]),
    "axvs_ffn_fwd": (C.c_int, [_fp, _fp, _fp, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int, _fp, C.c_size_t, _fp]),
    "axvs_cc_layer_packed_bytes": (C.c_size_t, []),
    "axvs_cc_layer_pack": (C.c_int, [C.POINTER(AxvsCCLayerParams), _fp, C.c_int, _fp]),
    "axvs_cc_layer_workspace_bytes": (C.c_size_t, [C.c_int] * 3),
    "axvs_cc_layer_fwd": (C.c_int, [_fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, _fp, C.c_size_t, _fp]),
    "axvs_cc_heads_packed_bytes": (C.c_size_t, [C.c_int]),
    "axvs_cc_heads_pack": (C.c_int, [C.POINTER(AxvsCCHeadParams), _fp, C.c_int, C.c_int, _fp]),
    "axvs_cc_heads_workspace_bytes": (C.c_size_t, [C.c_int] * 3),
    "axvs_cc_heads_fwd": (C.c_int, [_fp, _fp, _fp, _fp, _fp] + [C.c_int] * 8 + [_fp, C.c_size_t, _fp]),
    "axvs_conv1x1_gn_packed_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "axvs_conv1x1_gn_pack": (C.c_int, [C.POINTER(AxvsConvGnParams), _fp, C.c_int, C.c_int, C.c_int, _fp]),
    "axvs_conv1x1_gn_workspace_bytes": (C.c_size_t, [C.c_int] * 4),
    "axvs_conv1x1_gn_fwd": (C.c_int, [_fp, C.c_int, C.c_longlong, C.c_longlong, _fp, C.c_int, C.c_longlong, C.c_longlong, _fp] +
                            [C.c_int] * 5 + [C.c_float, C.c_int, _fp, C.c_size_t, _fp]),
    "axvs_conv1x1_gn_train_saved_bytes": (C.c_size_t, [C.c_int] * 6 + [C.c_longlong, C.c_longlong]),
    "axvs_conv1x1_gn_train_scratch_bytes": (C.c_size_t, [C.c_int] * 6),
    "axvs_conv1x1_gn_train_fwd": (C.c_int, [_fp, C.c_int, C.c_longlong, C.c_longlong, _fp, C.c_int, C.c_longlong, C.c_longlong, C.POINTER(AxvsConvGnParams)] +
                                  [C.c_int] * 5 + [C.c_float, _fp, C.c_size_t, _fp, C.c_size_t, _fp]),
    "axvs_conv1x1_gn_train_bwd": (C.c_int, [_fp, C.c_int, C.c_longlong, C.c_longlong, _fp, C.c_int, C.c_longlong, C.c_longlong, C.POINTER(AxvsConvGnParams),
                                            C.POINTER(AxvsConvGnParams), _fp] + [C.c_int] * 5 + [_fp, C.c_size_t, _fp, C.c_size_t, _fp]),
    "axvs_linear_sum_assignment": (C.c_int, [_fp, _fp, C.c_int, C.c_int, _fp]),
    "axvs_match_embds_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "axvs_match_embds": (C.c_int, [_fp, _fp, _fp, C.c_int, C.c_int, _fp, C.c_size_t, _fp]),
    "axvs_match_clips_workspace_bytes": (C.c_size_t, [C.c_int] * 4),
    "axvs_match_clips": (C.c_int, [_fp, _fp] + [C.c_int] * 4 + [_fp, C.c_size_t, _fp]),
    "axvs_add_channel_vector": (C.c_int, [_fp, _fp, C.c_size_t, C.c_int, _fp]),
    "axvs_pos2d": (C.c_int, [_fp, _fp] + [C.c_int] * 4 + [C.c_longlong, C.c_longlong, C.c_float, C.c_int, C.c_float, _fp]),
    "axvs_msda_packed_bytes": (C.c_size_t, [C.c_int] * 4),
    "axvs_msda_pack": (C.c_int, [C.POINTER(AxvsMsdaParams), _fp] + [C.c_int] * 5 + [_fp]),
    "axvs_msda_workspace_bytes": (C.c_size_t, [C.c_int] * 7),
    "axvs_msda_fwd": (C.c_int, [_fp, _fp, C.c_int, _fp, _fp, C.POINTER(C.c_int), _fp, _fp] + [C.c_int] * 8 + [_fp, C.c_size_t, _fp]),
    "axvs_msda_sample_fwd": (C.c_int, [_fp, _fp, _fp, C.c_int, _fp, _fp, C.POINTER(C.c_int), _fp, _fp] + [C.c_int] * 8 + [_fp, C.c_size_t, _fp]),
    "axvs_msda_output_proj_fwd": (C.c_int, [_fp, _fp, _fp, _fp, C.c_longlong] + [C.c_int] * 5 + [_fp]),
    "axvs_msda_layer_packed_bytes": (C.c_size_t, [C.c_int] * 5),
    "axvs_msda_layer_pack": (C.c_int, [C.POINTER(AxvsMsdaLayerParams), _fp] + [C.c_int] * 6 + [_fp]),
    "axvs_msda_layer_workspace_bytes": (C.c_size_t, [C.c_int] * 7),
    "axvs_msda_layer_fwd": (C.c_int, [_fp, _fp, _fp, C.c_int, _fp, C.POINTER(C.c_int), _fp, _fp] + [C.c_int] * 8 + [_fp, C.c_size_t, _fp]),
    "axvs_msda_core_fwd": (C.c_int, [_fp, C.POINTER(C.c_int), _fp, _fp, _fp] + [C.c_int] * 7 + [_fp]),
    "axvs_msda_core_bwd": (C.c_int, [_fp, C.POINTER(C.c_int), _fp, _fp, _fp, _fp, _fp, _fp] + [C.c_int] * 7 + [_fp]),
    "axvs_cc_module_workspace_bytes": (C.c_size_t, [C.c_int] * 4),
    "axvs_cc_module_fwd": (C.c_int, [_fp] * 5 + [C.POINTER(_fp), _fp] + [C.c_int] * 8 + [C.POINTER(C.c_int), C.c_int, _fp, C.c_size_t, _fp]),
    "axvs_tl_cc_module_workspace_bytes": (C.c_size_t, [C.c_int] * 5),
    "axvs_tl_cc_module_fwd": (C.c_int, [_fp] * 5 + [C.POINTER(_fp), _fp] + [C.c_int] * 9 + [C.POINTER(C.c_int), C.c_int, _fp, C.c_size_t, _fp]),
    "axvs_tl_heads_packed_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "axvs_tl_heads_pack": (C.c_int, [C.POINTER(AxvsTLHeadParams), _fp, C.c_int, C.c_int, C.c_int, _fp]),
    "axvs_tl_heads_workspace_bytes": (C.c_size_t, [C.c_int] * 4),
    "axvs_tl_heads_fwd": (C.c_int, [_fp, _fp, _fp, _fp, _fp] + [C.c_int] * 9 + [_fp, C.c_size_t, _fp]),
    "axvs_pos3d": (C.c_int, [_fp] + [C.c_int] * 5 + [C.c_float, C.c_int, C.c_float, _fp]),
    "axvs_pos3d_masked": (C.c_int, [_fp, _fp] + [C.c_int] * 5 + [C.c_float, C.c_int, C.c_float, _fp]),
    "axvs_scaled_residual": (C.c_int, [_fp, _fp, _fp, _fp, C.c_size_t, C.c_int, _fp]),
}

_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """Load libaxvs.so (once) and attach the prototypes.  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"axial_vs_amd: {LIB_PATH} is missing -- the HIP extension has not been built and there is no "
                "fallback path.  Run `python -c 'import __graft_entry__ as g; g.build()'` in the repo root.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


ERR_STATE = -4          # include/axvs.h AXVS_ERR_STATE: an earlier merged launch reported a hand-off timeout
_state_handler = None   # set by axial_vs_amd.modules: puts the sync words / status word back in order before the error is raised


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().axvs_last_error().decode(errors="replace")
        if rc == ERR_STATE and _state_handler is not None:
            _state_handler()
        raise RuntimeError(f"axial_vs_amd: {what} failed (code {rc}): {msg}")


# ---- torch.autocast -> 16-bit products in the training tier -------------------------------------------------------------------------
# Under autocast the reference's nn.Linear layers multiply 16-bit operands with fp32 accumulation.  The training tier does the same
# when it is called under autocast: library option "train_amp" (1: bf16 pieces, 2: fp16 pieces, one per operand) for the duration
# of the forward call, and again for the backward call of the same graph.  `AMP_COMPUTE = False` (or `module.amp_compute = False`)
# keeps the split-precision (fp32-accurate) products under autocast.
AMP_COMPUTE = True
_AMP_MODE = 0


def autocast_mode(owner=None) -> int:
    import torch
    if not torch.is_autocast_enabled() or not AMP_COMPUTE or (owner is not None and not getattr(owner, "amp_compute", True)):
        return 0
    return 2 if torch.get_autocast_dtype("cuda") == torch.float16 else 1


def current_amp() -> int:
    return _AMP_MODE


_AMP_LOCK = threading.RLock()


class train_amp:
    """Context manager: the library's X W^T GEMMs of the training tier run with `mode` (0 off, 1 bf16, 2 fp16) inside.
    The option is process-wide in the library (forward and backward threads must agree on it), so training-tier calls are
    serialised on a lock while one holds the option: a concurrent call from another thread (a second model, a DataParallel
    replica) waits instead of running its GEMMs at the other call's precision."""

    def __init__(self, mode: int):
        self.mode = int(mode)

    def __enter__(self):
        global _AMP_MODE
        _AMP_LOCK.acquire()
        self.prev = _AMP_MODE
        _AMP_MODE = self.mode
        try:
            if self.mode != self.prev:
                check(lib().axvs_set_option(b"train_amp", self.mode), "axvs_set_option")
        except BaseException:       # __exit__ will not run: put the mode back and let the other threads in
            _AMP_MODE = self.prev
            _AMP_LOCK.release()
            raise
        return self

    def __exit__(self, *exc):
        global _AMP_MODE
        _AMP_MODE = self.prev
        try:
            if self.mode != self.prev:
                check(lib().axvs_set_option(b"train_amp", self.prev), "axvs_set_option")
        finally:
            _AMP_LOCK.release()
        return False
