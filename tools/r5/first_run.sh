#!/bin/bash
# round-5 baseline on one box: pinned-status probe, clean kernel stats at the metric shape, PMC traffic at config 5's share,
# the shipped shapes (two-launch form today), full bench line
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r5_first; mkdir -p $O
python3 tools/r5/probe_pinned_status.py > $O/probe_pinned.txt 2>&1
bash tools/kstats.sh r5_base > $O/kstats_base.txt 2>&1
bash tools/kstats.sh r5_base_cfg5 --shape 8,4,256,96,96 --steps 40 > $O/kstats_cfg5.txt 2>&1
for s in 1,2,256,49,85 1,2,256,48,80 1,2,256,25,43 1,2,256,24,40 1,5,256,64,64 1,4,256,64,64 1,5,256,32,32 1,4,256,32,32 1,5,256,25,43 1,5,256,49,85; do
  python3 bench.py --shape $s --steps 200 --no-extras --no-cpu-baseline --no-qkav 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$s', d['ms_per_step'] * 1e3, 'us  frac', r['frac'], r['stage_us'])" >> $O/shapes.txt 2>&1
done
bash tools/pmc_traffic.sh --shape 8,4,256,96,96 _cfg5 > $O/pmc_cfg5.txt 2>&1
bash tools/pmc_traffic.sh > $O/pmc_metric.txt 2>&1
python3 bench.py > $O/bench_full.json 2> $O/bench_full.err
tail -c 600 $O/probe_pinned.txt; cat $O/kstats_base.txt $O/kstats_cfg5.txt $O/shapes.txt; tail -5 $O/pmc_cfg5.txt
