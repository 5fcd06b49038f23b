#!/bin/bash
# round-5 final evidence on ONE box: full GPU suite, clean kernel stats (timed region only) at the metric shape and at config 5's share,
# PMC traffic at both, the full bench line
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r5_final; mkdir -p $O
timeout 1800 python3 -m pytest tests -m gpu -q > $O/pytest.txt 2>&1
echo "pytest rc=$?" >> $O/pytest.txt; tail -4 $O/pytest.txt
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
bash tools/kstats.sh r5_final > $O/kstats.txt 2>&1
bash tools/kstats.sh r5_final_cfg5 --shape 8,4,256,96,96 --steps 40 > $O/kstats_cfg5.txt 2>&1
bash tools/kstats.sh r5_final_vipseg --shape 1,2,256,49,85 > $O/kstats_vipseg.txt 2>&1
bash tools/kstats.sh r5_final_tubelink --shape 1,5,256,24,40 > $O/kstats_tl.txt 2>&1
cat $O/kstats.txt $O/kstats_cfg5.txt $O/kstats_vipseg.txt $O/kstats_tl.txt | grep -v "^W2026"
bash tools/pmc_traffic.sh > $O/pmc_metric.txt 2>&1; tail -4 $O/pmc_metric.txt
bash tools/pmc_traffic.sh --shape 1,2,256,49,85 _vipseg > $O/pmc_vipseg.txt 2>&1; tail -4 $O/pmc_vipseg.txt
python3 bench.py > $O/bench_full.json 2> $O/bench_full.err
python3 - <<PY
import json
d = json.load(open("$O/bench_full.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["launch_us"], d["roofline"]["traffic"], d["roofline"]["qk_av"]["frac"], d["cpu_baseline"])
for k, v in d["extras"].items():
    print(k, json.dumps({a: b for a, b in v.items() if a != "what"})[:600])
PY
