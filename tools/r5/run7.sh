#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r5_run7; mkdir -p $O
timeout 600 python3 -m pytest tests/test_hip_parity.py -m gpu -q -x -k "axial_layer_golden or ffn or merged_qkv_launch or cfg5 or shard" > $O/pytest.txt 2>&1
echo "pytest rc=$?" >> $O/pytest.txt; tail -5 $O/pytest.txt
bash tools/ab_run.sh 3 > $O/ab_metric.txt 2>&1
tail -16 $O/ab_metric.txt
for so in uni0 uni1 uni0 uni1; do
  AXVS_LIB_PATH=$R/tools/ab/$so.so python3 bench.py --shape 8,4,256,96,96 --steps 40 --no-extras --no-cpu-baseline --no-qkav 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$so cfg5', d['ms_per_step'] * 1e3, d['roofline']['frac'], d['roofline']['stage_us'])" >> $O/ab_cfg5.txt 2>&1
done
cat $O/ab_cfg5.txt
