#!/bin/bash
# per-kernel average durations of the bench workload: tools/kstats.sh <tag> [bench args...]  -> gpurun_out/kstats_<tag>/
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=$1; shift
OUT=$R/gpurun_out/kstats_$tag
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline "$@" > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-220
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/*/*kernel_stats.csv"):
    rows = [r for r in csv.DictReader(open(f)) if "axvs" in r["Name"] and int(r["Calls"]) >= 100]
    tot = 0.0
    per_step = min(int(r["Calls"]) for r in rows)          # the least-launched kernel runs once per forward (settling steps included)
    for r in sorted(rows, key=lambda r: -float(r["AverageNs"])):
        n = round(int(r["Calls"]) / per_step)
        tot += float(r["AverageNs"]) * n
        print(f"  {r['Name'][:70]:70s} calls/step {n}  avg {float(r['AverageNs'])/1e3:8.2f} us")
    print(f"  kernel sum per step: {tot/1e3:.2f} us")
    import shutil; shutil.copy(f, "$OUT/kernel_stats.csv")
PY
