#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for o in 4 2 4 2; do
timeout 600 python3 bench.py --no-cpu-baseline --no-qkav --opt msda_gemm=$o 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        e = json.loads(l)['extras']; print('msda_gemm=$o: cfg3', e['wc_cfg3']['ms_per_forward'], 'vipseg module', e['wc_cfg3']['ms_per_forward_vipseg_r50_769x1345_T2'])
"; done
