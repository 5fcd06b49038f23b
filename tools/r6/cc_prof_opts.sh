#!/bin/bash
# per-kernel durations of the cross-clip module at BASELINE config 4 with library options: tools/r6/cc_prof_opts.sh <tag> [key=value ...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
OUT=$R/gpurun_out/ccprof_$tag
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/cc_time.py "$@" > $OUT/log.txt 2>&1
grep "us per forward" $OUT/log.txt | head -2
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/*/*kernel_stats.csv"):
    rows = [r for r in csv.DictReader(open(f)) if int(r["Calls"]) >= 50]
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:9]:
        print(f"  {r['Name'][:90]:90s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.2f} us")
    import shutil; shutil.copy(f, "$OUT/kernel_stats.csv")
PY
