#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/ccvideo
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/cc_video.py > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/*/*kernel_stats.csv"):
    rows = [r for r in csv.DictReader(open(f))]
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
        print(f"  {r['Name'][:100]:100s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.2f} us  {100*float(r['TotalDurationNs'])/tot:5.1f} %")
PY
