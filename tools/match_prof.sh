cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/matchprof
mkdir -p $OUT
python3 $R/tools/match_time.py | tail -3
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/match_time.py > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/*/*kernel_stats.csv"):
    for r in sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))[:8]:
        print(f"  {r['Name'][:90]:90s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.2f} us min {float(r['MinNs'])/1e3:8.2f}")
PY
