#!/bin/bash
# per-kernel durations of the within-clip module at BASELINE config 3: tools/cfg3_prof.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/cfg3prof_$1
mkdir -p $OUT
python3 $R/tools/cfg3_time.py 30 | tail -1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/cfg3_time.py 30 > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/*/*kernel_stats.csv"):
    rows = [r for r in csv.DictReader(open(f)) if int(r["Calls"]) >= 30]
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:24]:
        print(f"  {r['Name'][:96]:96s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.2f} us  {100*float(r['TotalDurationNs'])/tot:5.1f} %")
    print("  total kernel ms", tot / 1e6)
    import shutil; shutil.copy(f, "$OUT/kernel_stats.csv")
PY
