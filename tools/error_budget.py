"""Error budget of the 16-bit operand roundings of one axial-trajectory layer (and of a stack of them), CPU only.

Every point where the HIP pipeline rounds a value to fp16 (MFMA operands, 16-bit intermediates in LDS / HBM) is a "site".  For each
site the layer is run in fp32 with ONLY that site rounded (its own contribution) and with ALL sites rounded EXCEPT that one (what
making it exact -- split precision: 3x the MFMA work of that GEMM -- would buy), against the float64 oracle.  A free-running stack of
layers shows how the per-layer error accumulates (the within-clip decoder runs 2-4 axial layers + 2 deformable layers per level).

    python tools/error_budget.py [--json profiles/r2_error_budget.json]
"""
import json, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
import axvs_oracle as orc

SITES = ["w_qkv", "w_temporal", "w_ffn", "a_qk", "a_v", "q", "k", "v", "p", "x", "q2", "qk", "xbar", "o", "y", "h"]
DESC = {"w_qkv": "q/k/v weights", "w_temporal": "proj_q / proj_kv / proj weights", "w_ffn": "linear1 / linear2 weights",
        "a_qk": "src + pos as the q/k projection operand", "a_v": "src as the v projection operand", "q": "q (16-bit, HBM)", "k": "k (16-bit, HBM)",
        "v": "V^T (16-bit, HBM)", "p": "exp2(logits - max) as the P operand", "x": "attention output x (16-bit, LDS)",
        "q2": "q2 as the operand of Wk2^T q2", "qk": "Wk2^T q2 as the fdot2 operand", "xbar": "sum_f a_f x_f accumulated in packed f16",
        "o": "o as the operand of proj", "y": "norm1 output as the linear1 operand", "h": "relu(linear1) as the linear2 operand"}


def f16(x):
    return x.to(torch.float16).float()


class Q:
    def __init__(self, on):
        self.on = set(on)

    def __call__(self, site, x):
        return f16(x) if site in self.on else x


def lin(x, w, name, q, wsite, bias=True):
    y = x @ q(wsite, w[name + ".weight"]).t()
    return y + w[name + ".bias"] if bias else y


def traj(kq, val, w, T, heads, q):
    S, N, C = kq.shape
    L, d = N // T, C // heads
    scale = d ** -0.5
    Qm = q("q", lin(q("a_qk", kq), w, "q", q, "w_qkv") * scale)
    K = q("k", lin(q("a_qk", kq), w, "k", q, "w_qkv"))
    V = q("v", lin(q("a_v", val), w, "v", q, "w_qkv"))
    sp = lambda t: t.reshape(S, N, heads, d).permute(0, 2, 1, 3)
    Qh, Kh, Vh = sp(Qm), sp(K), sp(V)
    x = torch.empty(S, N, T, C)
    for f in range(T):
        lg = Qh @ Kh[:, :, f * L:(f + 1) * L].transpose(-1, -2)
        e = torch.exp(lg - lg.max(-1, keepdim=True).values)
        xf = (q("p", e) @ Vh[:, :, f * L:(f + 1) * L]) / e.sum(-1, keepdim=True)
        x[:, :, f] = xf.permute(0, 2, 1, 3).reshape(S, N, C)
    x = q("x", x)
    own = torch.arange(N) // L
    xd = x[:, torch.arange(N), own]
    q2 = q("q2", lin(xd, w, "proj_q", q, "w_temporal") * scale).reshape(S, N, heads, d)
    Wkv = q("w_temporal", w["proj_kv.weight"])
    Wk2, Wv2 = Wkv[:C].reshape(heads, d, C), Wkv[C:]
    qk = q("qk", torch.einsum("snhd,hdc->snhc", q2, Wk2))                     # (Wk2_h^T q2_h): the k2 bias drops out of the softmax
    tl = torch.einsum("snhc,sntc->snth", qk, x)
    ta = torch.softmax(tl, 2)                                                  # [S,N,T,heads]
    xbar = torch.zeros(S, N, heads, C)
    for f in range(T):                                                          # packed-f16 running sum (v_pk_fma_f16)
        xbar = q("xbar", xbar + q("xbar", ta[:, :, f]).unsqueeze(-1) * x[:, :, f].unsqueeze(2))
    o = torch.einsum("snhc,hdc->snhd", xbar, Wv2.reshape(heads, d, C)).reshape(S, N, C) + w["proj_kv.bias"][C:]
    return lin(q("o", o), w, "proj", q, "w_temporal")


def layer(src, pos, w, heads, q):
    B, T, H, W, C = pos.shape
    x = src.reshape(B, T, H, W, C)
    xs = x.permute(0, 3, 1, 2, 4).reshape(B * W, T * H, C)
    ps = pos.permute(0, 3, 1, 2, 4).reshape(B * W, T * H, C)
    xs = xs + traj(xs + ps, xs, orc._sub(w, "height_attn"), T, heads, q)
    x = xs.reshape(B, W, T, H, C).permute(0, 2, 3, 1, 4)
    xs = x.permute(0, 2, 1, 3, 4).reshape(B * H, T * W, C)
    ps = pos.permute(0, 2, 1, 3, 4).reshape(B * H, T * W, C)
    xs = xs + traj(xs + ps, xs, orc._sub(w, "width_attn"), T, heads, q)
    x = xs.reshape(B, H, T, W, C).permute(0, 2, 1, 3, 4).reshape(B * T, H * W, C)
    x = orc._layer_norm(x, w, "norm1")
    ff = lin(q("h", torch.relu(lin(q("y", x), w, "linear1", q, "w_ffn"))), w, "linear2", q, "w_ffn")
    return orc._layer_norm(x + ff, w, "norm2")


def errs(y, ref):
    e = y.double() - ref
    return float(e.abs().max() / ref.abs().max()), float(e.norm() / ref.norm())


if __name__ == "__main__":
    B, T, C, H, W, F = 1, 4, 256, 32, 32, 1024
    torch.manual_seed(0)
    w = orc.random_weights(orc.axial_layer_param_shapes(C, F), 0)
    src, pos = orc.synthetic_clip(B, T, C, H, W, 0)
    ref, _, _ = orc.axial_layer(src.double(), pos.double(), w, 8, want_attn=False)
    out = {"workload": f"one TemporalAxialTrajectoryAttentionLayer, [B={B},T={T},C={C},H={H},W={W}], d_ffn {F}, fp16 roundings on the CPU",
           "metric": "(max|a-b| / max|b|, relative L2) against the float64 oracle", "sites": {}}
    all_on = errs(layer(src, pos, w, 8, Q(SITES)), ref)
    none = errs(layer(src, pos, w, 8, Q([])), ref)
    out["all_sites_fp16"], out["no_site_rounded_fp32"] = all_on, none
    print(f"all sites fp16: max/max {all_on[0]:.2e} relL2 {all_on[1]:.2e};   fp32 everywhere: {none[0]:.1e} / {none[1]:.1e}")
    print(f"{'site':11s} {'alone: max/max':>15s} {'relL2':>9s} | {'all but it: max/max':>20s} {'relL2':>9s}   what")
    for s in SITES:
        a = errs(layer(src, pos, w, 8, Q([s])), ref)
        b = errs(layer(src, pos, w, 8, Q([t for t in SITES if t != s])), ref)
        out["sites"][s] = {"what": DESC[s], "alone": a, "all_but": b}
        print(f"{s:11s} {a[0]:15.2e} {a[1]:9.2e} | {b[0]:20.2e} {b[1]:9.2e}   {DESC[s]}")
    # free-running stack: the output of layer i (fp16 pipeline) feeds layer i+1; reference: the same stack in float64
    ws = [orc.random_weights(orc.axial_layer_param_shapes(C, F), 10 + i) for i in range(6)]
    x16, x64 = src, src.double()
    out["stack"] = []
    for i, wi in enumerate(ws):
        x16 = layer(x16, pos, wi, 8, Q(SITES))
        x64, _, _ = orc.axial_layer(x64, pos.double(), wi, 8, want_attn=False)
        e = errs(x16, x64)
        out["stack"].append({"layers": i + 1, "max_over_max": e[0], "rel_l2": e[1]})
        print(f"stack of {i + 1} layer(s), free-running: max/max {e[0]:.2e} relL2 {e[1]:.2e}")
    # round 5 (review item 6): the same stack with the two activation sites of a "split" mode made exact (src + pos into q / k,
    # norm1 output into linear1), and with the activation AND weight sites of those two GEMMs exact, over three weight seeds
    if "--split" in sys.argv:
        out["split_stack"] = []
        variants = {"all_fp16": SITES, "split_a_qk_y": [t for t in SITES if t not in ("a_qk", "y")],
                    "split_a_qk_y_h": [t for t in SITES if t not in ("a_qk", "y", "h")],
                    "split_a_qk_y_h_and_their_weights": [t for t in SITES if t not in ("a_qk", "y", "h", "w_qkv", "w_ffn")]}
        for seed in (10, 20, 30):
            ws = [orc.random_weights(orc.axial_layer_param_shapes(C, F), seed + i) for i in range(6)]
            x64 = src.double()
            for wi in ws:
                x64, _, _ = orc.axial_layer(x64, pos.double(), wi, 8, want_attn=False)
            for name, on in variants.items():
                x16 = src
                for wi in ws:
                    x16 = layer(x16, pos, wi, 8, Q(on))
                e = errs(x16, x64)
                out["split_stack"].append({"seed": seed, "variant": name, "max_over_max": e[0], "rel_l2": e[1]})
                print(f"6-layer stack, seed {seed}, {name:34s}: max/max {e[0]:.2e} relL2 {e[1]:.2e}")
    if "--json" in sys.argv:
        json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
