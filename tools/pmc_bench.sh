#!/bin/bash
# rocprofv3 PMC passes over bench.py (tuning helper); per-kernel counter averages
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pmc_bench
mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC SQ_INSTS_SALU" "GRBM_GUI_ACTIVE GRBM_COUNT" ; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/set$i -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-qkav --no-stages --settle-ms 0 "$@" > $OUT/set$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in sorted(glob.glob("$OUT/set*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void axvs::", "")[:48]
        if "at::" in k or "rocclr" in k or "pack" in k: continue
        a = agg[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
import json
summary = {}
for k, cs in agg.items():
    summary[k] = {c: round(v / n, 1) for c, (v, n) in sorted(cs.items())}
    w = summary[k].get("SQ_WAVE_CYCLES"); g = summary[k].get("GRBM_GUI_ACTIVE")
    if w:
        # SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD
        # (MI355X_MICROARCH.md, per-instruction constants): fractions below are of the same unit
        summary[k]["derived"] = {
            "valu_active_frac_of_wave_cycles": round(summary[k].get("SQ_ACTIVE_INST_VALU", 0) / w, 4),
            "wait_inst_any_frac_of_wave_cycles": round(summary[k].get("SQ_WAIT_INST_ANY", 0) / w, 4),
            "lds_active_frac_of_wave_cycles": round(summary[k].get("SQ_ACTIVE_INST_LDS", 0) / w, 4),
            "lds_bank_conflict_frac_of_lds_active": round(summary[k].get("SQ_LDS_BANK_CONFLICT", 0) / max(summary[k].get("SQ_LDS_IDX_ACTIVE", 1), 1), 4),
            # kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs; MFMA busy is summed over the chip's 1024 SIMDs
            "mfma_busy_frac": round(summary[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(g / 8 * 1024, 1), 4) if g else None,
        }
json.dump({"command": "rocprofv3 --pmc <set> --kernel-trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --settle-ms 0 (3 passes, one counter set each)",
           "per_kernel_per_launch": summary}, open("$OUT/summary.json", "w"), indent=1)
for k, cs in agg.items():
    print(k)
    w = cs.get("SQ_WAVE_CYCLES", [0, 1]); wc = w[0] / max(w[1], 1)
    for c, (v, n) in sorted(cs.items()):
        x = v / n
        pct = f"  ({100 * x / wc:5.1f}% of WAVE_CYCLES)" if wc and c.startswith("SQ_") and c not in ("SQ_WAVE_CYCLES",) else ""
        print(f"    {c:28s} {x:14.0f}{pct}")
PY
