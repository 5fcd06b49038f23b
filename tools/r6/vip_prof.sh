#!/bin/bash
# per-kernel durations of the within-clip module at the shipped VIPSeg ResNet-50 setting: tools/r6/vip_prof.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/vipprof_$1
rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/r6/vip_module_time.py 1 > $OUT/log.txt 2>&1
tail -2 $OUT/log.txt
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/*/*kernel_stats.csv"):
    rows = [r for r in csv.DictReader(open(f)) if int(r["Calls"]) >= 45]
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:26]:
        print(f"  {r['Name'][:110]:110s} calls/fwd {int(r['Calls'])/45:5.1f} avg {float(r['AverageNs'])/1e3:8.2f} us  {100*float(r['TotalDurationNs'])/tot:5.1f} %")
    print("  kernel sum per forward (us)", tot / 45 / 1e3)
    import shutil; shutil.copy(f, "$OUT/kernel_stats.csv")
PY
rm -rf $OUT/*/
