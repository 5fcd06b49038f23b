#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r5_plan
timeout 900 python3 -m pytest tests/test_hip_parity.py -m gpu -x -q -k "small_problem or two_chunks or ragged or gelu or ffn or graphed or two_streams" 2>&1 | tail -3
timeout 600 python3 tools/r5/opt_sweep.py ffn_split_finish 1,0 1,4,16,16 1,4,32,32 1,2,25,43 1,5,12,20 1,5,24,40 1,8,16,16 2>&1 | grep -v amdgpu | cut -c1-150 | tee gpurun_out/r5_plan/ffn_fin.txt
for o in 0 1 0 1; do
timeout 600 python3 bench.py --no-cpu-baseline --no-qkav --opt ffn_split_finish=$o 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        e = json.loads(l)['extras']; print('ffn_split_finish=$o: cfg3', e['wc_cfg3']['ms_per_forward'], 'vipseg module', e['wc_cfg3']['ms_per_forward_vipseg_r50_769x1345_T2'], 'tl 24x40', e['tl_t5']['[1,5,256,24,40]']['us_per_layer'], 'vip 25x43', e['vipseg_t2']['[1,2,256,25,43]']['us_per_layer'])
"; done
