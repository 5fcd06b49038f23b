#!/bin/bash
# per-kernel durations of a training step at BASELINE config 2: tools/train_prof.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/trainprof_$1
mkdir -p $OUT
python3 $R/tools/train_time.py 10 | tail -2
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/train_time.py 10 > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/*/*kernel_stats.csv"):
    rows = [r for r in csv.DictReader(open(f))]
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:22]:
        print(f"  {r['Name'][:90]:90s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.2f} us  {100*float(r['TotalDurationNs'])/tot:5.1f} %")
    import shutil; shutil.copy(f, "$OUT/kernel_stats.csv")
PY
