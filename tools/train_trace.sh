#!/bin/bash
# per (kernel, grid) durations of the layer's training step at BASELINE config 2: which launches are slow
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/traintrace
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/tools/train_time.py 5 "$@" > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/*/*kernel_trace.csv")[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    key = (r["Kernel_Name"][:70], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"])
    d[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = sorted(d.items(), key=lambda kv: -sum(kv[1]))
tot = sum(sum(v) for v in d.values())
for k, v in rows[:40]:
    v2 = sorted(v)
    print(f"{k[0]:70s} grid {k[1]:>8s}x{k[2]:>3s}x{k[3]:>3s} calls {len(v):5d} med {v2[len(v2)//2]:8.2f} min {v2[0]:8.2f} share {100*sum(v)/tot:5.1f} %")
PY
rm -rf $OUT/*/
