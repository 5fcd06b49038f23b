#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r5_run11; mkdir -p $O
timeout 900 python3 -m pytest tests/test_hip_parity.py -m gpu -q -x -k "golden or ffn or merged_qkv_launch or cfg5 or shard or ragged or output_map or encoder or within_clip or decoder" > $O/pytest.txt 2>&1
echo "pytest rc=$?" >> $O/pytest.txt; tail -6 $O/pytest.txt
bash tools/ab_run.sh 3 > $O/ab_metric.txt 2>&1
tail -16 $O/ab_metric.txt
for so in one0 one1 one0 one1; do
  AXVS_LIB_PATH=$R/tools/ab/$so.so python3 bench.py --shape 8,4,256,96,96 --steps 40 --no-extras --no-cpu-baseline --no-qkav 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$so cfg5', d['ms_per_step'] * 1e3, d['roofline']['frac'], d['roofline']['stage_us'])" >> $O/ab2.txt 2>&1
  AXVS_LIB_PATH=$R/tools/ab/$so.so python3 bench.py --shape 2,4,256,64,64 --steps 200 --no-extras --no-cpu-baseline --no-qkav 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$so cfg2', d['ms_per_step'] * 1e3, d['roofline']['frac'])" >> $O/ab2.txt 2>&1
  echo "$so ffn standalone 16384 rows: $(AXVS_LIB_PATH=$R/tools/ab/$so.so python3 tools/ffn_time.py 16384 2>/dev/null | tail -1)" >> $O/ab2.txt
  echo "$so ffn standalone 4096 rows: $(AXVS_LIB_PATH=$R/tools/ab/$so.so python3 tools/ffn_time.py 4096 2>/dev/null | tail -1)" >> $O/ab2.txt
done
cat $O/ab2.txt
