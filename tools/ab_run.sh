#!/bin/bash
# tools/ab_run.sh [rounds] -- per-kernel averages (rocprofv3 --kernel-trace --stats) of bench.py for every tools/ab/*.so, interleaved
# `rounds` times on this box; summary in gpurun_out/ab_summary.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
ROUNDS=${1:-2}
shift
OUT=$R/gpurun_out/ab
rm -rf $OUT; mkdir -p $OUT
for r in $(seq 1 $ROUNDS); do
  for so in $R/tools/ab/*.so; do
    n=$(basename $so .so)
    AXVS_LIB_PATH=$so rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${n}_$r -- python3 $R/bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-extras --no-qkav --no-stages "$@" > $OUT/${n}_$r.log 2>&1
    # averages over the timed region only (the last 300 launches of every layer kernel in the raw trace)
    python3 $R/tools/trace_reduce.py $OUT/${n}_$r $OUT/${n}_$r.log --out $OUT/${n}_$r.stats.csv > /dev/null; rm -rf $OUT/${n}_$r
  done
done
python3 - <<PY > $R/gpurun_out/ab_summary.txt
import csv, glob, collections, json, os
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$OUT/*.stats.csv")):
    n = os.path.basename(f)[:-len(".stats.csv")].rsplit("_", 1)[0]
    rows = [r for r in csv.DictReader(l for l in open(f) if not l.startswith("#"))]
    for r in rows:
        k = r["Name"].replace("void axvs::", "").split("(")[0][:52]
        res[n][k].append(float(r["AverageNs"]) / 1e3 * int(r["CallsPerStep"]))
for f in sorted(glob.glob("$OUT/*.log")):
    n = os.path.basename(f).rsplit("_", 1)[0]
    for l in open(f):
        if l.startswith("{"):
            res[n]["ms_per_step(us)"].append(json.loads(l)["ms_per_step"] * 1e3)
kernels = sorted({k for v in res.values() for k in v})
for k in kernels:
    print(k)
    for n in sorted(res):
        v = res[n].get(k, [])
        if v: print(f"    {n:24s} " + " ".join(f"{x:8.2f}" for x in v) + f"   | min {min(v):8.2f}")
print("kernel sum per step (min over rounds):")
for n in sorted(res):
    print(f"    {n:24s} {sum(min(v) for k, v in res[n].items() if k != 'ms_per_step(us)'):8.2f}")
PY
cat $R/gpurun_out/ab_summary.txt
