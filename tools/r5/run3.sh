#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r5_run3; mkdir -p $O
timeout 1800 python3 -m pytest tests -m gpu -q -x --deselect tests/test_hip_parity.py::test_graphed_forwards_own_their_sync_words > $O/pytest.txt 2>&1
echo "pytest rc=$?" >> $O/pytest.txt
tail -30 $O/pytest.txt
timeout 300 python3 -m pytest tests/test_hip_parity.py -m gpu -q -k "graphed_forwards_own or ragged_frames_in or hand_off_timeout or two_streams_with_grids or range_report or output_map_in_16" > $O/pytest_new.txt 2>&1
tail -30 $O/pytest_new.txt
for s in 1,2,256,49,85 1,2,256,48,80 1,2,256,25,43 1,4,256,49,85 1,4,256,64,64; do
  python3 bench.py --shape $s --steps 200 --no-extras --no-cpu-baseline --no-qkav 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$s', d['ms_per_step'] * 1e3, 'us  frac', r['frac'], r['stage_us'])" >> $O/shapes.txt 2>&1
done
cat $O/shapes.txt
