#!/bin/bash
# round-6 evidence on ONE box: smoke, clean kernel stats (timed region only) at the metric shape / config 5's share / shipped shapes, per-kernel stats of
# BASELINE configs 3 and 4, PMC traffic at the metric shape, the full bench line.  (The GPU suite runs separately: tools/grun.sh ... pytest -m gpu.)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r6_final; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
bash tools/kstats.sh r6_final > $O/kstats.txt 2>&1
bash tools/kstats.sh r6_final_cfg5 --shape 8,4,256,96,96 --steps 40 > $O/kstats_cfg5.txt 2>&1
bash tools/kstats.sh r6_final_vipseg --shape 1,2,256,49,85 > $O/kstats_vipseg.txt 2>&1
cat $O/kstats.txt $O/kstats_cfg5.txt $O/kstats_vipseg.txt | grep -v "^W2026\|^E2026"
bash tools/cfg3_prof.sh r6 > $O/cfg3_prof.txt 2>&1; grep -v "^W2026\|^E2026" $O/cfg3_prof.txt | tail -30
bash tools/cc_prof.sh r6 > $O/cc_prof.txt 2>&1; grep -v "^W2026\|^E2026" $O/cc_prof.txt | tail -14
bash tools/pmc_traffic.sh > $O/pmc_metric.txt 2>&1; tail -4 $O/pmc_metric.txt
python3 bench.py > $O/bench_full.json 2> $O/bench_full.err
python3 - <<PY
import json
d = json.load(open("$O/bench_full.json"))
rf = d["roofline"]
print(d["value"], d["ms_per_step"], rf["frac"], rf.get("frac_events"), rf["launch_us"], rf["traffic"], rf["qk_av"]["frac"], d["cpu_baseline"])
print(json.dumps(rf.get("kernels")))
for k, v in d["extras"].items():
    print(k, json.dumps({a: b for a, b in v.items() if a != "what"})[:900])
PY
