#!/bin/bash
# L2 hit rate per kernel (tuning helper)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pmc_l2
mkdir -p $OUT
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_RDREQ_sum" "TCC_TAG_STALL_sum TCC_BUSY_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/set$i -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --settle-ms 0 > $OUT/set$i.log 2>&1
  tail -2 $OUT/set$i.log | cut -c1-200
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in sorted(glob.glob("$OUT/set*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void axvs::", "")[:48]
        if "at::" in k or "rocclr" in k or "pack" in k: continue
        a = agg[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, cs in agg.items():
    print(k)
    for c, (v, n) in sorted(cs.items()):
        print(f"    {c:34s} {v / n:16.0f}")
    h, m = cs.get("TCC_HIT_sum", [0, 1]), cs.get("TCC_MISS_sum", [0, 1])
    if h[0] + m[0] > 0: print(f"    L2 hit rate {h[0] / (h[0] + m[0]):.3f}")
PY
