#!/bin/bash
# per-kernel durations of a training step of the within-clip module at BASELINE config 3: tools/wc_train_prof.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/wctrainprof_$1
mkdir -p $OUT
python3 $R/tools/wc_train_time.py 10
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/wc_train_time.py 10 > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/*/*kernel_stats.csv"):
    rows = [r for r in csv.DictReader(open(f))]
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("total kernel time", tot / 1e6, "ms over the run (26 steps + 13 forwards)")
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:26]:
        print(f"  {r['Name'][:100]:100s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.2f} us  {100*float(r['TotalDurationNs'])/tot:5.1f} %")
    import shutil; shutil.copy(f, "$OUT/kernel_stats.csv")
PY
rm -rf $OUT/*/
