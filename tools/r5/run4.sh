#!/bin/bash
# wave-specialised FFN: parity first (the default build is WS), then same-box A/B against the lockstep body
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r5_run4; mkdir -p $O
timeout 900 python3 -m pytest tests/test_hip_parity.py -m gpu -q -x -k "axial_layer_golden or cfg5 or merged_qkv or ffn or output_map or ragged or batch or shard or determin or encoder_golden" > $O/pytest.txt 2>&1
echo "pytest rc=$?" >> $O/pytest.txt
tail -12 $O/pytest.txt
bash tools/ab_run.sh 2 > $O/ab_metric.txt 2>&1
tail -25 $O/ab_metric.txt
for so in ws0 ws1 ws0 ws1; do
  AXVS_LIB_PATH=$R/tools/ab/$so.so python3 bench.py --shape 8,4,256,96,96 --steps 40 --no-extras --no-cpu-baseline --no-qkav 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$so cfg5', d['ms_per_step'] * 1e3, d['roofline']['frac'], d['roofline']['stage_us'])" >> $O/ab_cfg5.txt 2>&1
  AXVS_LIB_PATH=$R/tools/ab/$so.so python3 bench.py --shape 2,4,256,64,64 --steps 200 --no-extras --no-cpu-baseline --no-qkav 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$so cfg2', d['ms_per_step'] * 1e3, d['roofline']['frac'], d['roofline']['stage_us'])" >> $O/ab_cfg5.txt 2>&1
done
cat $O/ab_cfg5.txt
