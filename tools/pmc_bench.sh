#!/bin/bash
# rocprofv3 PMC passes over bench.py (tuning helper); per-kernel counter averages
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_bench
mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC SQ_INSTS_SALU" "GRBM_GUI_ACTIVE GRBM_COUNT" ; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/set$i -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/set$i.log 2>&1
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
    w = cs.get("SQ_WAVE_CYCLES", [0, 1]); wc = w[0] / max(w[1], 1)
    for c, (v, n) in sorted(cs.items()):
        x = v / n
        pct = f"  ({100 * x / wc:5.1f}% of WAVE_CYCLES)" if wc and c.startswith("SQ_") and c not in ("SQ_WAVE_CYCLES",) else ""
        print(f"    {c:28s} {x:14.0f}{pct}")
PY
