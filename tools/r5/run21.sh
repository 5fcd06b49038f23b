#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r5_plan
i=0
for o in "" "--no-qkav" "--no-cpu-baseline"; do
i=$((i+1))
timeout 900 python3 bench.py $o > gpurun_out/r5_plan/benchf_$i.json 2> gpurun_out/r5_plan/benchf_$i.err
python3 - $i "$o" <<'PY'
import json,sys
d=json.loads([l for l in open(f'gpurun_out/r5_plan/benchf_{sys.argv[1]}.json') if l.startswith('{')][-1])
e=d['extras']['wc_cfg3']
print(sys.argv[2] or 'full', d['ms_per_step'], e['ms_per_forward'], e['ms_per_forward_vipseg_r50_769x1345_T2'], e['ms_per_forward_f32_stack'])
PY
done
