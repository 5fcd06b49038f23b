#!/bin/bash
# per-kernel durations of the cross-clip module at BASELINE config 4: tools/cc_prof.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/ccprof_$1
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/cc_time.py > $OUT/log.txt 2>&1
tail -3 $OUT/log.txt
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/*/*kernel_stats.csv"):
    rows = [r for r in csv.DictReader(open(f)) if int(r["Calls"]) >= 50]
    tot = 0.0
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
        print(f"  {r['Name'][:100]:100s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.2f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
        tot += float(r["TotalDurationNs"])
    print("  total kernel time ms", tot / 1e6)
    import shutil; shutil.copy(f, "$OUT/kernel_stats.csv")
PY
