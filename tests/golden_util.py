"""Helpers to replay tests/golden/*.npz (made by oracle/gen_golden.py from the reference)."""
import json
import os

import numpy as np
import torch

import axvs_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    meta["shapes"] = {k: tuple(v) for k, v in meta["shapes"].items()}
    return z, meta


def weights(z, meta, seed=None):
    """Regenerate the fixture's weights from its seed; verify against the stored checksum."""
    w = orc.random_weights(meta["shapes"], meta["seed"] if seed is None else seed)
    got = float(sum(v.double().sum() for v in w.values()))
    assert abs(got - float(z["wsum"])) < 1e-6 * max(1.0, abs(got)), "torch RNG drift: regenerate fixtures"
    return w


def t(a):
    return torch.from_numpy(np.asarray(a))


def rel_err(a, b):
    """max |a-b| / max |b|  (the parity metric used throughout: error relative to the tensor's scale)."""
    a = a.double()
    b = b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def rel_l2(a, b):
    a = a.double()
    b = b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def checks(x):
    x = x.double()
    return np.array([x.sum().item(), (x * x).sum().item(), x.abs().max().item()])


def axial_inputs(meta):
    g = torch.Generator().manual_seed(meta["seed"] + 1)
    B, T, C, H, W = (meta[k] for k in "BTCHW")
    x = torch.randn(B, T, C, H, W, generator=g)
    src = x.permute(0, 1, 3, 4, 2).reshape(B * T, H * W, C).contiguous()
    pos = orc.pos_embed_sine_3d(B, T, H, W, C // 2)
    return src, pos


AXIAL = ["g2_axial_B1_T2_C128_H32_W32", "g2_axial_B2_T3_C64_H5_W7", "g2_axial_B1_T5_C64_H6_W4",
         "g2_axial_B1_T1_C64_H4_W5", "g2_axial_B1_T4_C256_H64_W64"]
TRAJ = ["g1_traj_S3_T2_L7_C64", "g1_traj_S2_T5_L6_C64", "g1_traj_S2_T1_L9_C64", "g1_traj_S4_T4_L16_C256"]
