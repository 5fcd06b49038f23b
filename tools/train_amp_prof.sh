#!/bin/bash
# per-kernel durations of ONE autocast(bf16) training step (kept activations) of the axial layer: tools/train_amp_prof.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/trainamp
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/tools/train_time.py 6 --amp > $OUT/log.txt 2>&1
tail -2 $OUT/log.txt
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last fwd+bwd step before the forward-only loop of the second configuration: find the last tr_ln_bwd (backward marker)
idx = [i for i, r in enumerate(rows) if "tr_ln_bwd" in r["Kernel_Name"]]
# a step has 2 tr_ln_bwd kernels (norm2, norm1): the step ends a few kernels after the last one; take the window between the
# 2nd-last pair's end and the last pair's end
end = idx[-1]
while end + 1 < len(rows) and int(rows[end + 1]["Start_Timestamp"]) - int(rows[end]["End_Timestamp"]) < 200000: end += 1
start = idx[-3]
while start + 1 < len(rows) and int(rows[start + 1]["Start_Timestamp"]) - int(rows[start]["End_Timestamp"]) < 200000 and start < idx[-2] - 5: start += 1
# simpler: one step = kernels after the previous step's last kernel; steps are separated by the python-side gap -> use 3 steps average
agg = collections.OrderedDict()
seg = rows[idx[-7] + 1: idx[-1] + 1] if len(idx) >= 7 else rows
for r in seg:
    k = r["Kernel_Name"].replace("void axvs::", "").replace("axvs::", "").split("(")[0][:70]
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
nsteps = 3
tot = sum(v[1] for v in agg.values())
print(f"kernels per step {sum(v[0] for v in agg.values()) / nsteps:.0f}, kernel time per step {tot / nsteps:.1f} us")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:70s} x{v[0] / nsteps:5.1f}  avg {v[1] / v[0]:8.2f} us  per step {v[1] / nsteps:8.1f} us")
PY
