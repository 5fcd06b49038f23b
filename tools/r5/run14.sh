#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r5_run14; mkdir -p $O
timeout 400 python3 tools/merge_soak.py 240 > $O/soak.txt 2>&1; echo "soak rc=$?" >> $O/soak.txt; tail -3 $O/soak.txt
python3 -c "import axial_vs_amd as ax; ax.check_status(); print('status clean')" >> $O/soak.txt 2>&1
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err ) 2> $O/time.txt
tail -3 $O/time.txt
python3 -c "
import json; d = json.load(open('$O/bench_driver.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], list(d['extras'].keys()))"
timeout 1500 python3 -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt; tail -3 $O/pytest.txt
