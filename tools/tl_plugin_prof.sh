#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/tlprof
mkdir -p $OUT
python3 $R/tools/tl_plugin_time.py 30 | tail -1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/tl_plugin_time.py 30 > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/*/*kernel_stats.csv"):
    rows = [r for r in csv.DictReader(open(f)) if int(r["Calls"]) >= 30]
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:16]:
        print(f"  {r['Name'][:96]:96s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.2f} us  {100*float(r['TotalDurationNs'])/tot:5.1f} %")
PY
