#!/bin/bash
# HBM traffic of one layer forward from the TCC counters (MI355X_MICROARCH.md "HBM"): separate --pmc passes for
# FETCH_SIZE and WRITE_SIZE; FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950 (corrected below).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
# usage: tools/pmc_traffic.sh [--shape B,T,C,H,W] [tag]   (default: the metric shape -> gpurun_out/pmc_traffic/traffic.json)
SHAPE=1,4,256,64,64
if [ "$1" = "--shape" ]; then SHAPE=$2; shift 2; fi
TAG=${1:-}
OUT=$R/gpurun_out/pmc_traffic$TAG
rm -rf $OUT; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-qkav --no-stages --settle-ms 0 --shape $SHAPE > $OUT/$c.log 2>&1
done
python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in sorted(glob.glob("$OUT/*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void axvs::", "")
        if "at::" in k or "rocclr" in k or "pack" in k or "pos3d" in k: continue
        a = agg[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
# launches per layer forward: the FFN-carrying trajectory kernel (or, failing that, the least-launched axvs kernel) runs once per forward
cnt = {k: max(c[1] for c in cs.values()) for k, cs in agg.items()}
nfwd = min(cnt.values())
tot_r = tot_w = 0.0
rows = {}
for k, cs in agg.items():
    fs = cs.get("FETCH_SIZE", [0, 1]); ws = cs.get("WRITE_SIZE", [0, 1])
    rd = fs[0] / max(fs[1], 1) * 1024 * 2      # KB -> bytes, x2 gfx950 correction for wide coalesced reads
    wr = ws[0] / max(ws[1], 1) * 1024
    n = round(cnt[k] / nfwd)
    rows[k[:60]] = {"read_MB_per_launch": round(rd / 1e6, 2), "write_MB_per_launch": round(wr / 1e6, 2), "launches_per_layer": n}
    tot_r += rd * n; tot_w += wr * n
res = {"per_kernel": rows, "layer_read_MB": round(tot_r / 1e6, 1), "layer_write_MB": round(tot_w / 1e6, 1),
       "layer_total_MB": round((tot_r + tot_w) / 1e6, 1),
       "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE x2 (gfx950: 128-B requests tallied at 64 B)"}
print(json.dumps(res, indent=1))
res["workload"] = "axial layer fwd [B,T,C,H,W] = [$SHAPE] d_ffn=1024 f16"
B_, T_, C_, H_, W_ = (int(v) for v in "$SHAPE".split(","))
res["algorithmic_MB"] = round((3 * B_ * T_ * H_ * W_ * C_ * 4 + 2 * (2 * (7 * C_ * C_ + 8 * C_) + 2 * C_ * 1024 + 1024 + 5 * C_)) / 1e6, 1)
json.dump(res, open("$OUT/traffic.json", "w"), indent=1)
PY
