#!/bin/bash
# per-kernel durations of the layer forward at one shape: tools/shape_prof.sh B T H W [option=value ...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/shapeprof
rm -rf $OUT; mkdir -p $OUT
cat > $OUT/run.py <<PY
import sys, os, torch
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/oracle")
import axvs_oracle as orc, axial_vs_amd as ax
from axial_vs_amd import _lib
B, T, H, W = [int(a) for a in sys.argv[1:5]]
for a in sys.argv[5:]:
    k, v = a.split("="); _lib.check(_lib.lib().axvs_set_option(k.encode(), int(v)), k)
C, F = 256, 1024
layer = ax.TemporalAxialTrajectoryAttentionLayer(C, F, n_heads=8).eval()
layer.load_state_dict(orc.random_weights(orc.axial_layer_param_shapes(C, F), 5), strict=True)
layer = layer.cuda()
pg = ax.PositionEmbeddingSine3D(C // 2, normalize=True).channels_last(B, T, H, W, "cuda")
src = torch.randn(B * T, H * W, C, device="cuda")
for _ in range(30): layer(src, pg)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100): layer(src, pg)
e1.record(); torch.cuda.synchronize()
print(f"layer [{B},{T},256,{H},{W}]: {e0.elapsed_time(e1) * 10:.1f} us per forward")
PY
python3 $OUT/run.py "$@" | tail -1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $OUT/run.py "$@" > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/*/*kernel_stats.csv"):
    rows = [r for r in csv.DictReader(open(f)) if int(r["Calls"]) >= 100]
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:8]:
        print(f"  {r['Name'][:96]:96s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.2f} us")
PY
rm -rf $OUT/*/
