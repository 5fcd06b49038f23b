#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r5_run13; mkdir -p $O
timeout 900 python3 -m pytest tests/test_hip_parity.py tests/test_hip_cc_training.py -m gpu -q -k "cross_clip or cc_ or tl_ or tube" > $O/pytest.txt 2>&1
echo "pytest rc=$?" >> $O/pytest.txt; tail -8 $O/pytest.txt
for o in "" "--opt no_cc_proj_fusion=1" "" "--opt no_cc_proj_fusion=1"; do
  python3 bench.py --workload cc --steps 100 $o 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('[$o]', d['ms_per_step'] * 1e3, 'us')" >> $O/cc_ab.txt
done
cat $O/cc_ab.txt
python3 bench.py --steps 50 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python3 -c "
import json; d = json.load(open('$O/bench.json')); print(d['value'], {k: v for k, v in d['extras']['cc_cfg4'].items() if k not in ('what', 'shape')}, d['extras']['wc_cfg3']['ms_per_forward'])"
