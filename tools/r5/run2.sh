#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r5_run2; mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1
echo "pytest rc=$?" >> $O/pytest.txt
tail -15 $O/pytest.txt
python3 bench.py --steps 100 > $O/bench.json 2> $O/bench.err
python3 - <<PY
import json
d = json.load(open("$O/bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
for k in ("cfg5_share", "cfg5_gather", "vipseg_t2", "tl_t5"):
    print(k, json.dumps({a: b for a, b in d["extras"][k].items() if a != "what"}))
PY
