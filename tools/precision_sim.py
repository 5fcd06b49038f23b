"""Scratch study: error of the axial layer vs the float64 oracle when MFMA operands are rounded to
bf16 / fp16 / split-bf16 at the points the HIP pipeline rounds them.  CPU only."""
import sys, os, math, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
import axvs_oracle as orc

def make_q(mode):
    if mode == "f32": return lambda x: x
    if mode == "bf16": return lambda x: x.to(torch.bfloat16).float()
    if mode == "f16": return lambda x: x.to(torch.float16).float()
    if mode == "bf16x2":  # hi+lo split: ~16 mantissa bits
        def q(x):
            hi = x.to(torch.bfloat16).float()
            lo = (x - hi).to(torch.bfloat16).float()
            return hi + lo
        return q
    raise ValueError(mode)

def lin(x, w, name, q):
    return q(x) @ q(w[name + ".weight"]).t() + w[name + ".bias"]

def traj(kq, val, w, T, heads, q, qs):
    S, N, C = kq.shape; L = N // T; d = C // heads; scale = d ** -0.5
    Q = q(lin(kq, w, "q", q)); K = q(lin(kq, w, "k", q)); V = q(lin(val, w, "v", q))
    Qh = Q.reshape(S, N, heads, d).permute(0, 2, 1, 3); Kh = K.reshape(S, N, heads, d).permute(0, 2, 1, 3)
    Vh = V.reshape(S, N, heads, d).permute(0, 2, 1, 3)
    x = torch.empty(S, N, T, C)
    for f in range(T):
        lg = Qh @ Kh[:, :, f*L:(f+1)*L].transpose(-1, -2) * scale
        p = torch.softmax(lg, -1)
        # kernel: P unnormalised in bf16, normalise after PV in fp32
        m = lg.max(-1, keepdim=True).values
        e = torch.exp(lg - m); s = e.sum(-1, keepdim=True)
        xf = (qs(e) @ Vh[:, :, f*L:(f+1)*L]) / s
        x[:, :, f] = xf.permute(0, 2, 1, 3).reshape(S, N, C)
    x = q(x)
    own = torch.arange(N) // L
    xd = x[:, torch.arange(N), own]
    q2 = lin(xd, w, "proj_q", q) * scale
    kv = lin(x, w, "proj_kv", q)
    k2, v2 = kv[..., :C], kv[..., C:]
    tl = (q2.reshape(S, N, 1, heads, d) * k2.reshape(S, N, T, heads, d)).sum(-1)
    ta = torch.softmax(tl, 2)
    o = (ta.unsqueeze(-1) * v2.reshape(S, N, T, heads, d)).sum(2).reshape(S, N, C)
    return lin(o, w, "proj", q)

def layer(src, pos, w, heads, mode, pmode=None):
    q = make_q(mode); qs = make_q(pmode or mode)
    B, T, H, W, C = pos.shape
    x = src.reshape(B, T, H, W, C)
    xs = x.permute(0, 3, 1, 2, 4).reshape(B * W, T * H, C); ps = pos.permute(0, 3, 1, 2, 4).reshape(B * W, T * H, C)
    xs = xs + traj(xs + ps, xs, orc._sub(w, "height_attn"), T, heads, q, qs)
    x = xs.reshape(B, W, T, H, C).permute(0, 2, 3, 1, 4)
    xs = x.permute(0, 2, 1, 3, 4).reshape(B * H, T * W, C); ps = pos.permute(0, 2, 1, 3, 4).reshape(B * H, T * W, C)
    xs = xs + traj(xs + ps, xs, orc._sub(w, "width_attn"), T, heads, q, qs)
    x = xs.reshape(B, H, T, W, C).permute(0, 2, 1, 3, 4).reshape(B * T, H * W, C)
    x = orc._layer_norm(x, w, "norm1")
    ff = lin(q(torch.relu(lin(x, w, "linear1", q))), w, "linear2", q)
    return orc._layer_norm(x + ff, w, "norm2")

if __name__ == "__main__":
    B, T, C, H, W = [int(a) for a in (sys.argv[1:6] or [1, 4, 256, 32, 32])]
    w = orc.random_weights(orc.axial_layer_param_shapes(C, 1024), 0)
    src, pos = orc.synthetic_clip(B, T, C, H, W, 0)
    ref, _, _ = orc.axial_layer(src.double(), pos.double(), w, 8, want_attn=False)
    ref32, _, _ = orc.axial_layer(src, pos, w, 8, want_attn=False)
    def rep(name, y):
        e = (y.double() - ref)
        print(f"{name:10s} max/max {float(e.abs().max()/ref.abs().max()):.2e}  relL2 {float(e.norm()/ref.norm()):.2e}  "
              f"max elementwise-rel(|ref|>0.1) {float((e.abs()/ref.abs())[ref.abs()>0.1].max()):.2e}")
    rep("torch f32", ref32)
    for mode in ["f32", "bf16", "f16", "bf16x2"]:
        rep(mode, layer(src, pos, w, 8, mode))
