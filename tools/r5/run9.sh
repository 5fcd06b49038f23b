#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r5_run9; mkdir -p $O
python3 - > $O/parity.txt 2>&1 <<'PY'
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "oracle")
import axial_vs_amd as ax
from axial_vs_amd import _lib
L = _lib.lib()
def names():
    return [L.axvs_profile_stage_name(i).decode() for i in range(L.axvs_profile_stage_count())]
for shape in [(8, 4, 96, 96), (4, 4, 96, 96), (6, 3, 48, 80), (5, 2, 49, 85)]:
    B, T, H, W = shape
    layer = ax.TemporalAxialTrajectoryAttentionLayer(256, 1024, n_heads=8).eval().cuda()
    pos = ax.PositionEmbeddingSine3D(128, normalize=True).channels_last(B, T, H, W, "cuda")
    ok = True
    for it in range(3):
        src = torch.randn(B * T, H * W, 256, device="cuda")
        one = layer(src, pos)[0].clone(); n1 = names()
        L.axvs_set_option(b"no_persist", 1)
        two = layer(src, pos)[0].clone(); n2 = names()
        L.axvs_set_option(b"no_persist", 0)
        ok = ok and torch.equal(one, two)
    torch.cuda.synchronize(); ax.check_status()
    print(shape, "bit-identical" if ok else "DIFFERENT", n1[1:], "vs", n2[1:])
PY
cat $O/parity.txt
for o in "" "--opt no_persist=1" "" "--opt no_persist=1"; do
  python3 bench.py --shape 8,4,256,96,96 --steps 40 --no-extras --no-cpu-baseline --no-qkav $o 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('[$o] cfg5', d['ms_per_step'] * 1e3, d['roofline']['frac'], d['roofline']['stage_us'])" >> $O/ab.txt 2>&1
  python3 bench.py --shape 4,4,256,96,96 --steps 80 --no-extras --no-cpu-baseline --no-qkav $o 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('[$o] B4', d['ms_per_step'] * 1e3, d['roofline']['frac'], d['roofline']['stage_us'])" >> $O/ab.txt 2>&1
done
cat $O/ab.txt
