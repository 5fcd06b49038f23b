#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r5_plan
for o in 128 65 128 65; do
timeout 600 python3 bench.py --no-cpu-baseline --no-qkav --opt small_tiles_below=$o > gpurun_out/r5_plan/bench_$o.json 2> gpurun_out/r5_plan/bench_$o.err
python3 - $o <<'PY'
import json,sys
d=json.loads([l for l in open(f'gpurun_out/r5_plan/bench_{sys.argv[1]}.json') if l.startswith('{')][-1])
e=d['extras']['wc_cfg3']
print(sys.argv[1], d['ms_per_step'], e['ms_per_forward'], e['ms_per_forward_vipseg_r50_769x1345_T2'], e['ms_per_forward_f32_stack'])
PY
done
