#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for o in 0 1 2 0 1 2; do
timeout 600 python3 bench.py --no-cpu-baseline --no-qkav --opt ffn_wide=$o 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        e = json.loads(l)['extras']; print('ffn_wide=$o: cfg3', e['wc_cfg3']['ms_per_forward'], 'vipseg module', e['wc_cfg3']['ms_per_forward_vipseg_r50_769x1345_T2'], 'tl_t5 64x64', e['tl_t5']['[1,5,256,64,64]']['us_per_layer'])
"; done
