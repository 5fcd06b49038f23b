#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r5_plan
for o in 0 64 128 1 0 64 128 1; do
timeout 600 python3 bench.py --no-cpu-baseline --no-qkav --opt merge_small=$o > gpurun_out/r5_plan/bench_ms$o.json 2> gpurun_out/r5_plan/bench_ms$o.err
python3 - $o <<'PY'
import json,sys
d=json.loads([l for l in open(f'gpurun_out/r5_plan/bench_ms{sys.argv[1]}.json') if l.startswith('{')][-1])
e=d['extras']
print('merge_small', sys.argv[1], 'headline', d['ms_per_step'], 'cfg3', e['wc_cfg3']['ms_per_forward'], 'vipseg module', e['wc_cfg3']['ms_per_forward_vipseg_r50_769x1345_T2'], 'cc', e['cc_cfg4']['us_per_forward'],
      'vip 25x43', e['vipseg_t2']['[1,2,256,25,43]']['us_per_layer'], 'tl 12x20', e['tl_t5']['[1,5,256,12,20]']['us_per_layer'])
PY
done 2>&1 | tee gpurun_out/r5_plan/merge_small_bench.txt
