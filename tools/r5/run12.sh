#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r5_run12; mkdir -p $O
timeout 600 python3 -m pytest tests/test_hip_parity.py -m gpu -q -k "cross_clip or cc_" > $O/pytest.txt 2>&1
echo "pytest rc=$?" >> $O/pytest.txt; tail -5 $O/pytest.txt
python3 bench.py --steps 50 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python3 -c "
import json; d = json.load(open('$O/bench.json')); print(d['value'], {k: v for k, v in d['extras']['cc_cfg4'].items() if k != 'what'})"

