#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r5_run10; mkdir -p $O
timeout 1800 python3 -m pytest tests -m gpu -q > $O/pytest.txt 2>&1
echo "pytest rc=$?" >> $O/pytest.txt
tail -15 $O/pytest.txt
for s in 1,5,256,12,20 1,5,256,24,40 1,2,256,25,43 1,2,256,49,85; do
  python3 bench.py --shape $s --steps 200 --no-extras --no-cpu-baseline --no-qkav 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$s', d['ms_per_step'] * 1e3, 'us  frac', r['frac'], r['stage_us'])" >> $O/shapes.txt 2>&1
done
cat $O/shapes.txt
