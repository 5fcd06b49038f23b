"""One-box A/B of the few-rows shapes for every tools/ab/*.so (each variant in its own process): layer as a HIP-graph replay + the cross-clip module."""
import sys, os, subprocess, json
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, R)
    import torch
    import axial_vs_amd as ax
    out = {}
    for sh in ["1,4,16,16", "1,4,32,32", "1,2,25,43", "1,2,24,40", "1,5,12,20", "1,4,24,40", "1,8,16,16", "2,4,16,32"]:
        B, T, H, W = [int(v) for v in sh.split(",")]
        layer = ax.TemporalAxialTrajectoryAttentionLayer(256, 1024, n_heads=8).eval().cuda()
        s = torch.randn(B * T, H * W, 256, device="cuda")
        p = ax.PositionEmbeddingSine3D(128, normalize=True).channels_last(B, T, H, W, "cuda")
        g = ax.GraphedForward(layer, s, p)
        for _ in range(30): g()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(300): g()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 300 * 1e3)
        out[sh] = round(best, 2)
    print(json.dumps(out))
    sys.exit(0)
import glob
res = {}
for rnd in range(2):
    for so in sorted(glob.glob(os.path.join(R, "tools", "ab", "*.so"))):
        n = os.path.basename(so)[:-3]
        env = dict(os.environ, AXVS_LIB_PATH=so)
        o = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True).stdout
        j = json.loads([l for l in o.splitlines() if l.startswith("{")][-1])
        c = subprocess.run([sys.executable, os.path.join(R, "bench.py"), "--workload", "cc", "--steps", "200"], env=env, capture_output=True, text=True).stdout
        j["cc_cfg4"] = json.loads([l for l in c.splitlines() if l.startswith("{")][-1])["ms_per_step"] * 1e3
        res.setdefault(n, []).append(j)
keys = list(next(iter(res.values()))[0].keys())
print(f"{'us per layer / forward':>24s} | " + " | ".join(f"{n:>16s}" for n in res))
for k in keys:
    print(f"{k:>24s} | " + " | ".join(f"{' / '.join(f'{r[k]:.2f}' for r in res[n]):>16s}" for n in res))
