#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2 3; do for v in base nt; do AXVS_LIB_PATH=tools/ab/$v.so timeout 300 python3 bench.py --workload cc --steps 300 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'): print('$v cc cfg4 us per forward', round(json.loads(l)['ms_per_step']*1e3,1))
"; done; done
