#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python3 -m pytest tests/test_hip_parity.py -m gpu -x -q -k "cross_clip or tube_link or cc_" 2>&1 | tail -3
for i in 1 2 3; do timeout 300 python3 bench.py --workload cc --steps 300 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'): print('cc cfg4 us per forward', round(json.loads(l)['ms_per_step']*1e3,1))
"; done
timeout 600 bash tools/cc_prof.sh r5end2 2>&1 | grep "class_head\|gemm64\|total kernel"
