#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r5_plan
timeout 2400 python3 -m pytest tests -m gpu -x -q > gpurun_out/r5_plan/gpu_tests.txt 2>&1; tail -5 gpurun_out/r5_plan/gpu_tests.txt
timeout 900 python3 bench.py > gpurun_out/r5_plan/bench.json 2> gpurun_out/r5_plan/bench.err; tail -c 600 gpurun_out/r5_plan/bench.err
python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r5_plan/bench.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
e=d['extras']
for k in ('vipseg_t2','tl_t5'):
    for s,v in e[k].items():
        if isinstance(v,dict): print(k, s, v['us_per_layer'], v['launches'])
print('wc_cfg3', e['wc_cfg3'])
PY
