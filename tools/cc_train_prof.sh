#!/bin/bash
# per-kernel durations of a training step of the cross-clip module at BASELINE config 4: tools/cc_train_prof.sh <tag> [cc_train_time.py args, e.g. --shape 128,12,2,193,337,4]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/cctrainprof_$1
shift
mkdir -p $OUT
python3 $R/tools/cc_train_time.py 10 "$@"
python3 $R/tools/cc_train_time.py 10 --torch "$@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/cc_train_time.py 10 "$@" > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/*/*kernel_stats.csv"):
    rows = [r for r in csv.DictReader(open(f))]
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("total kernel time", tot / 1e6, "ms over the run")
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
        print(f"  {r['Name'][:90]:90s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.2f} us  {100*float(r['TotalDurationNs'])/tot:5.1f} %")
    import shutil; shutil.copy(f, "$OUT/kernel_stats.csv")
PY
rm -rf $OUT/*/
