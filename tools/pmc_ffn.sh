#!/bin/bash
# rocprofv3 PMC pass over the standalone FFN kernel (tuning helper); output in gpurun_out/pmc_ffn
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out/pmc_ffn
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT" ; do
  tag=$(echo $set | cut -d' ' -f1)
  AXVS_DET=1 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_ffn/$tag -- python3 $R/tools/ffn_unit.py 16384 5 > $R/gpurun_out/pmc_ffn/$tag.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$R/gpurun_out/pmc_ffn/*/*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: [0, 0])
    for r in csv.DictReader(open(f)):
        if "ffn_fused" not in r["Kernel_Name"]: continue
        a = agg[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, (v, n) in agg.items(): print(f"{k:28s} {v/n:16.1f}  per dispatch ({n} dispatches)")
PY
