"""CPU: the C-ABI library loads and exports every symbol include/axvs.h declares; host-side module surface."""
import ctypes
import os
import re

import pytest
import torch

import __graft_entry__ as ge
from golden_util import AXIAL, TRAJ, load

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="session", autouse=True)
def built():
    ge.build()


def header_symbols():
    text = open(os.path.join(ROOT, "include", "axvs.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(axvs_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from axial_vs_amd import _lib
    syms = header_symbols()
    assert len(syms) >= 10
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(raw, s), f"{s} declared in include/axvs.h but not exported"
    assert sorted(_lib.SIGNATURES) == syms, "ctypes prototypes out of sync with include/axvs.h"
    assert _lib.lib().axvs_version() >= 1


def test_size_queries_need_no_gpu():
    from axial_vs_amd import _lib
    L = _lib.lib()
    C, h, F = 256, 8, 1024
    packed = L.axvs_axial_layer_packed_bytes(C, h, F)
    # 2 x (7 C^2) 16-bit attention weights (+ 3 C^2 of per-head copies of the proj_kv halves for the reassociated temporal forms)
    # + 2 C F 16-bit FFN weights, plus fp32 biases / norms
    assert packed >= 2 * (2 * 10 * C * C) + 2 * (2 * C * F)
    assert packed < 1.1 * (2 * (2 * 10 * C * C) + 2 * (2 * C * F)) + 65536
    ws = L.axvs_axial_layer_workspace_bytes(1, 4, 64, 64, C, h, F)
    assert ws > 0
    assert L.axvs_traj_attn_workspace_bytes(4, 4, 16, C, h) > 0


def test_argument_errors_are_reported_not_raised_across_the_boundary():
    from axial_vs_amd import _lib
    L = _lib.lib()
    rc = L.axvs_axial_layer_fwd(None, None, None, None, 1, 4, 8, 8, 256, 8, 1024, 0, None, 0, None, None, None)
    assert rc == -1 and b"null" in L.axvs_last_error()
    rc = L.axvs_traj_pack(None, None, 250, 8, 0, None)
    assert rc == -1


@pytest.mark.parametrize("name", AXIAL[:2] + ["g4_encoder_B2_T2_C64_H6_W5"])
def test_state_dict_keys_match_reference(name):
    """strict load of a state dict with exactly the reference's parameter names and shapes."""
    import axial_vs_amd as ax
    import axvs_oracle as orc
    z, m = load(name)
    w = orc.random_weights(m["shapes"], 1)
    if "layers" in m:
        mod = ax.TemporalEncoder(m["C"], m["d_ffn"], n_heads=8, temporal_attn_type="axial-trajectory", num_temporal_layer=m["layers"])
    else:
        mod = ax.TemporalAxialTrajectoryAttentionLayer(m["C"], m["d_ffn"], n_heads=8)
    mod.load_state_dict(w, strict=True)
    assert set(mod.state_dict().keys()) == set(m["shapes"].keys())


def test_traj_state_dict_keys_match_reference():
    import axial_vs_amd as ax
    import axvs_oracle as orc
    z, m = load(TRAJ[0])
    mod = ax.TrajectoryAttention(m["C"], 8)
    mod.load_state_dict(orc.random_weights(m["shapes"], 1), strict=True)


def test_no_cpu_fallback_and_forward_only():
    import axial_vs_amd as ax
    layer = ax.TemporalAxialTrajectoryAttentionLayer(64, 128, n_heads=8)
    src, pos = torch.zeros(2, 12, 64), torch.zeros(1, 2, 3, 4, 64)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        layer(src, pos)                       # training mode: the training tier (tests/test_hip_training.py), GPU only as well
    with pytest.raises(NotImplementedError):
        ax.TemporalTrajectoryAttentionLayer(64, 128, n_heads=8)(src, pos)     # the full T*H*W layer has no training tier
    layer.eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        layer(src, pos)
    with pytest.raises(RuntimeError):
        ax.TemporalAxialTrajectoryAttentionLayer(64, 128, activation="swish")
    enc = ax.TemporalEncoder(64, 128, temporal_attn_type="axial_trajectory")   # the reference's default-string gotcha
    assert not hasattr(enc, "temporal_layers")
    full = ax.TemporalEncoder(64, 128, temporal_attn_type="trajectory", num_temporal_layer=1).eval()    # full T*H*W attention (a7)
    assert sorted(k for k in full.state_dict() if k.endswith(".q.weight")) == ["temporal_layers.0.temporal_attn.q.weight"]
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        full(src, pos)


@pytest.mark.parametrize("name", ["g5_cc_module_Q16_Tc3_V2_H8_L2", "g16_cc_module_syncbn_Q16_Tc3_V2_H8_L2"])
def test_cross_clip_state_dict_keys_match_reference(name):
    import axial_vs_amd as ax
    import axvs_oracle as orc
    z, m = load(name)
    mod = ax.CrossClipTrackingModule(num_layers=m["layers"], num_classes=m["num_classes"], attn_drop=0.0, aspp_drop=0.0,
                                     kernel_sizes=[3, 3, 3], atrous_rates=[1, 2, 3], norm_fn=m.get("norm_fn", "ln"), num_clip_frames=m["V"])
    ref_keys = set(m["shapes"].keys())
    own = {k: tuple(v.shape) for k, v in mod.state_dict().items() if v.dtype.is_floating_point}
    assert set(own.keys()) == ref_keys
    assert all(own[k] == m["shapes"][k] for k in ref_keys)
    sd = mod.state_dict()
    sd.update(orc.random_weights(m["shapes"], 1))
    mod.load_state_dict(sd, strict=True)


@pytest.mark.parametrize("name", ["g6_tl_cc_head_Tc3_Q16_f2_L2"])
def test_tube_link_head_state_dict_keys_match_reference(name):
    """TubeLinkCrossClipHead exposes the cross-clip members of the reference head under the reference's key names."""
    import axial_vs_amd as ax
    import axvs_oracle as orc
    z, m = load(name)
    mod = ax.TubeLinkCrossClipHead(num_classes=m["num_classes"], out_channels=m["Cm"], num_cc_layers=m["layers"])
    own = {k: tuple(v.shape) for k, v in mod.state_dict().items() if v.dtype.is_floating_point}
    assert set(own.keys()) == set(m["shapes"].keys())
    assert all(own[k] == tuple(m["shapes"][k]) for k in own)
    mod.load_state_dict(orc.random_weights(m["shapes"], 1), strict=True)
    with pytest.raises(RuntimeError):
        mod.eval()(torch.zeros(1, 3, 16, 256), torch.zeros(1, 6, m["Cm"], 8, 12))      # CPU tensors: no fallback


def test_msda_module_surface_matches_reference():
    """MSDeformAttn keeps the reference's parameter names / shapes / init (ops/modules/ms_deform_attn.py:59-79) and refuses
    CPU tensors (no fallback to a PyTorch sampling path)."""
    import axial_vs_amd as ax
    from golden_util import MSDA_MODULE
    z, m = load(MSDA_MODULE[0])
    mod = ax.MSDeformAttn(d_model=m["C"], n_levels=len(m["shapes"]), n_heads=m["M"], n_points=m["P"])
    own = {k: tuple(v.shape) for k, v in mod.state_dict().items()}
    assert own == {k: tuple(v) for k, v in m["wshapes"].items()}
    assert float(mod.sampling_offsets.weight.abs().max()) == 0.0 and float(mod.attention_weights.bias.abs().max()) == 0.0
    b = mod.sampling_offsets.bias.view(m["M"], len(m["shapes"]), m["P"], 2)
    assert torch.allclose(b[0, 0, :, 0], torch.arange(1, m["P"] + 1, dtype=torch.float32))     # head 0 points along +x, radius 1..P
    assert torch.allclose(b[:, :, 1], 2 * b[:, :, 0])
    with pytest.raises(RuntimeError):
        mod.eval()(torch.zeros(1, 4, m["C"]), torch.zeros(1, 4, len(m["shapes"]), 2), torch.zeros(1, 252, m["C"]), m["shapes"])


def _decoder_from_meta(m, cross_clip_training=False):
    import axial_vs_amd as ax

    class Shape:
        def __init__(self, c, s):
            self.channels, self.stride = c, s

    strides = {"res3": 8, "res4": 16, "res5": 32}
    return ax.WithinClipTrackingModule(
        {k: Shape(c, strides[k]) for k, c in m["chans"].items()}, transformer_dropout=0.0, transformer_attn_drop=0.0,
        transformer_nheads=8, transformer_dim_feedforward=m["d_ffn"], transformer_num_stages=m["stages"],
        transformer_spatial_layers=0 if m.get("temporal_only") else m["stages"], transformer_temporal_layers=m["stages"] * m["temporal_per_stage"],
        transformer_temporal_attn_type="axial-trajectory", transformer_conv_dims=256,
        transformer_spatial_in_features=["res3", "res4", "res5"], transformer_temporal_in_features=["res4", "res5"],
        num_clip_frames=m["T"], cross_clip_training=cross_clip_training)


def test_within_clip_module_state_dict_matches_reference():
    """WithinClipTrackingModule (pixel decoder: input/output projections, level embeddings, spatial + temporal stages) has the
    reference MSDeformAttnPixelDecoder's parameter names and shapes; CPU inputs raise."""
    import axvs_oracle as orc
    z, m = load("g8_pixel_decoder_T2_S2")
    mod = _decoder_from_meta(m)
    own = {k: tuple(v.shape) for k, v in mod.within_clip_tracking_module.state_dict().items()}
    assert own == {k: tuple(v) for k, v in m["shapes"].items()}
    mod.within_clip_tracking_module.load_state_dict(orc.random_weights(m["shapes"], 1), strict=True)
    with pytest.raises(RuntimeError):
        mod.eval().forward_features({k: torch.zeros(m["T"], c, *m["sizes"][k]) for k, c in m["chans"].items()})


def test_packed_weight_key_sees_replaced_parameters_and_modules():
    """The packed-weight cache key must change at once when a Parameter object is replaced by assignment, when it is updated
    in place, and when a whole submodule is replaced (the old key re-walked the module only every 256 calls)."""
    import axial_vs_amd as ax
    from axial_vs_amd.modules import _param_key
    layer = ax.TemporalAxialTrajectoryAttentionLayer(64, 128, n_heads=8)
    k0 = _param_key(layer, "f16")
    assert _param_key(layer, "f16") == k0 and _param_key(layer, "bf16") != k0
    layer.linear1.bias = torch.nn.Parameter(torch.zeros(128))                 # replaced Parameter object
    k1 = _param_key(layer, "f16")
    assert k1 != k0
    with torch.no_grad():
        layer.height_attn.q.weight.add_(1.0)                                  # in-place update (version counter)
    k2 = _param_key(layer, "f16")
    assert k2 != k1
    layer.norm2 = torch.nn.LayerNorm(64)                                      # replaced submodule
    k3 = _param_key(layer, "f16")
    assert k3 != k2 and _param_key(layer, "f16") == k3
    layer.load_state_dict(layer.state_dict())                                 # copy_ into the same Parameters
    assert _param_key(layer, "f16") != k3


def test_training_tier_has_no_cpu_fallback():
    """train() mode routes to the training tier, which refuses CPU tensors instead of falling back to torch."""
    import axial_vs_amd as ax
    import axvs_oracle as orc
    layer = ax.TemporalAxialTrajectoryAttentionLayer(64, 128, dropout=0.1, attn_drop=0.1, n_heads=8).train()
    src, pos = orc.synthetic_clip(1, 2, 64, 4, 4, 1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        layer(src, pos)
    from axial_vs_amd.training import layer_parameters
    names = [k for k, _ in layer.named_parameters()]
    assert len(layer_parameters(layer)) == len(names) == 32        # AxvsAxialLayerParams field order covers every parameter


def test_training_buffer_sizes_and_argument_checks():
    from axial_vs_amd import _lib
    L = _lib.lib()
    saved = L.axvs_axial_layer_train_saved_bytes(1, 4, 64, 64, 256, 8, 1024)
    M, C, T, F = 4 * 64 * 64, 256, 4, 1024
    assert saved >= 4 * (2 * (6 + 3 * T) * M * C + 4 * M * C + M * F)          # per pass q,k,v,xd,q2,o + x + kv2; pass outputs, z, u; r
    assert L.axvs_axial_layer_train_scratch_bytes(1, 4, 64, 64, 256, 8, 1024, 1) > L.axvs_axial_layer_train_scratch_bytes(1, 4, 64, 64, 256, 8, 1024, 0)
    assert L.axvs_axial_layer_train_saved_bytes(1, 4, 8, 8, 512, 4, 1024) == 0 and b"head_dim" in L.axvs_last_error()
    assert L.axvs_axial_layer_train_saved_bytes(1, 17, 8, 8, 256, 8, 1024) == 0 and b"T=17" in L.axvs_last_error()


def test_reference_extension_stand_in_exposes_the_two_entry_points():
    """`import MultiScaleDeformableAttention as MSDA` (OPS/functions/ms_deform_attn_func.py:22) resolves to the stand-in module and
    finds ms_deform_attn_forward / ms_deform_attn_backward with the extension's argument lists (OPS/src/ms_deform_attn.h:24-67)."""
    import importlib
    import inspect
    import sys
    import axial_vs_amd
    sys.path.insert(0, os.path.join(os.path.dirname(axial_vs_amd.__file__), "compat"))
    try:
        MSDA = importlib.import_module("MultiScaleDeformableAttention")
    finally:
        sys.path.pop(0)
    f = list(inspect.signature(MSDA.ms_deform_attn_forward).parameters)
    b = list(inspect.signature(MSDA.ms_deform_attn_backward).parameters)
    assert f == ["value", "value_spatial_shapes", "value_level_start_index", "sampling_locations", "attention_weights", "im2col_step"]
    assert b == ["value", "value_spatial_shapes", "value_level_start_index", "sampling_locations", "attention_weights", "grad_output", "im2col_step"]


def test_default_dtype_f32_falls_back_to_16_bit_where_there_is_no_fp32_tier():
    """`set_default_dtype("f32")` selects the fp32 tier of the axial layer; modules without such a tier (TrajectoryAttention, the
    cross-clip module, deformable attention, ...) keep 16-bit operands instead of failing on an unknown dtype (round-3 advisor)."""
    import axial_vs_amd as ax
    from axial_vs_amd import modules
    prev = modules._DEFAULT_DTYPE
    try:
        modules.set_default_dtype("f32")
        assert modules.default_operand_dtype() == "f16"
        assert ax.TrajectoryAttention(64, 8)._dtype() == "f16"
        layer = ax.TemporalAxialTrajectoryAttentionLayer(64, 128, n_heads=8)
        assert layer._dtype() == "f32"
        with pytest.raises(NotImplementedError):
            layer._pack()                      # the fp32 tier has no packed weights / per-pass entry point
        modules.set_default_dtype("bf16")
        assert modules.default_operand_dtype() == "bf16" and layer._dtype() == "bf16"
        with pytest.raises(ValueError):
            modules.set_default_dtype("fp8")
    finally:
        modules.set_default_dtype(prev)


def test_stack_precision_setter_reaches_every_axial_layer():
    """`WithinClipTrackingModule.set_stack_precision`: the operand precision of the axial-trajectory layers of the stack (the fp32
    setting is what holds 1e-3 in max-norm over the free-running stack of BASELINE config 3; GPU test
    test_within_clip_module_full_size_fp32_stack_holds_the_bar_in_max_norm)."""
    import axial_vs_amd as ax
    z, m = load("g8_pixel_decoder_T2_S2")
    mod = _decoder_from_meta(m)
    layers = [x for x in mod.modules() if isinstance(x, ax.TemporalAxialTrajectoryAttentionLayer)]
    assert layers and all(l.mfma_dtype is None for l in layers)
    assert mod.set_stack_precision("f32") is mod and all(l._dtype() == "f32" for l in layers)
    mod.set_stack_precision("f16")
    assert all(l._dtype() == "f16" for l in layers)
    with pytest.raises(ValueError):
        mod.set_stack_precision("fp8")


def test_forward_hooks_keep_the_split_and_cat_data_flow():
    """The decoder's eval path runs the temporal levels in place in the token buffer, bypassing `TemporalEncoder.forward()`; a
    module with forward (pre-)hooks must keep the reference's data flow, where the hooks see the per-level tensors."""
    import axial_vs_amd as ax
    from axial_vs_amd.modules import _has_hooks
    enc = ax.TemporalEncoder(256, 1024, temporal_attn_type="axial-trajectory", num_temporal_layer=2).eval()
    assert not _has_hooks(enc) and not any(_has_hooks(l) for l in enc.temporal_layers)
    h = enc.register_forward_hook(lambda m, a, o: None)
    assert _has_hooks(enc) and not enc.can_run_in_place(torch.zeros(1, 2, 4, 4, 256))
    h.remove()
    assert not _has_hooks(enc)
    h = enc.temporal_layers[1].register_forward_pre_hook(lambda m, a: None)
    assert _has_hooks(enc.temporal_layers[1])
    h.remove()
    assert not enc.train().can_run_in_place(torch.zeros(1, 2, 4, 4, 256))      # train() mode: autograd needs the out-of-place path


def test_status_check_and_state_error_are_part_of_the_abi():
    """Round 5: AXVS_ERR_STATE / axvs_check_status / option sync_spin_limit exist; without a registered word the check passes, and
    registering NULL (no GPU call involved) keeps it that way."""
    from axial_vs_amd import _lib
    L = _lib.lib()
    hdr = open(os.path.join(ROOT, "include", "axvs.h")).read()
    assert "#define AXVS_ERR_STATE (-4)" in hdr and _lib.ERR_STATE == -4
    assert L.axvs_set_status_buffer(None) == 0
    assert L.axvs_check_status() == 0
    assert L.axvs_set_option(b"sync_spin_limit", 1000) == 0 and L.axvs_set_option(b"sync_spin_limit", 0) == 0


def test_train_amp_context_releases_its_lock_when_the_option_call_fails(monkeypatch):
    """advisor (round 4): __enter__ took the lock and set the mode before a failing option call; __exit__ never ran."""
    from axial_vs_amd import _lib

    def boom(rc, what):
        raise RuntimeError("option call failed")
    prev = _lib.current_amp()
    monkeypatch.setattr(_lib, "check", boom)
    with pytest.raises(RuntimeError):
        with _lib.train_amp(1 if prev != 1 else 2):
            pass
    monkeypatch.undo()
    assert _lib.current_amp() == prev
    assert _lib._AMP_LOCK.acquire(blocking=False)       # free again (an RLock held by this thread would also succeed: check the count)
    _lib._AMP_LOCK.release()
    import threading
    got = []
    def other():
        ok = _lib._AMP_LOCK.acquire(timeout=2)
        got.append(ok)
        if ok:
            _lib._AMP_LOCK.release()
    t = threading.Thread(target=other)
    t.start(); t.join()
    assert got == [True], "another thread must be able to take the lock"


def test_in_place_levels_fall_back_when_the_strided_workspace_explodes():
    """advisor (round 4): a small level inside a long token buffer (B*T = 32 frames of 21504 tokens, a 32 x 32 level) would take
    ~680 MB per row-addressed temporary in place against 32 MB in the natural layout: can_run_in_place says no, the split / cat
    data flow runs; BASELINE config 3's stride is fine."""
    import axial_vs_amd as ax
    from axial_vs_amd import modules
    layer = ax.TemporalAxialTrajectoryAttentionLayer(256, 1024, n_heads=8).eval()
    pos_small = modules.tag_sine3d(torch.zeros(1, 4, 32, 32, 256), 10000.0, True, 6.283185307179586)
    assert layer.can_run_in_place(pos_small) and layer.can_run_in_place(pos_small, frame_stride_rows=64 * 64 + 32 * 32 + 16 * 16)
    pos_many = modules.tag_sine3d(torch.zeros(8, 4, 32, 32, 256), 10000.0, True, 6.283185307179586)
    assert not layer.can_run_in_place(pos_many, frame_stride_rows=21504)
    assert not layer.can_run_in_place(pos_many, frame_stride_rows=2 ** 22)      # row indices beyond 32 bits / 64
