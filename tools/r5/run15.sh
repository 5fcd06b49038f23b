#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r5_run15; mkdir -p $O
timeout 600 python3 -m pytest tests/test_hip_parity.py -m gpu -q -k "shipped_map_sizes" > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt; tail -4 $O/pytest.txt
python3 bench.py --steps 100 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
tail -1 $O/bench.json | python3 -c "
import sys, json; d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k: v for k, v in d['extras']['wc_cfg3'].items() if k != 'what'})"
tail -3 $O/bench.err
for o in "" "--opt no_merge_qkv=1"; do :; done
