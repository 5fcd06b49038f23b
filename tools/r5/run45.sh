#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2; do for o in 128 256 512; do timeout 300 python3 bench.py --workload cc --steps 300 --opt gemm_small_upto=$o 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'): print('gemm_small_upto=$o cc cfg4 us per forward', round(json.loads(l)['ms_per_step']*1e3,1))
"; done; done
for o in 128 256 512; do
timeout 600 python3 bench.py --no-cpu-baseline --no-qkav --opt gemm_small_upto=$o 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        e = json.loads(l)['extras']; print('gemm_small_upto=$o cfg3', e['wc_cfg3']['ms_per_forward'], 'vipseg module', e['wc_cfg3']['ms_per_forward_vipseg_r50_769x1345_T2'], 'cc', e['cc_cfg4']['us_per_forward'], 'train', e['train_step']['ms_per_step'], 'cc_train', e['cc_train_cfg4']['ms_per_step'])
"; done
