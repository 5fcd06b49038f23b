#!/bin/bash
# cross-clip module A/B over library options (same box, interleaved): tools/r6/cc_ab.sh "<opt list A>" "<opt list B>" ...   (each list: space-separated key=value)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "cross_clip or tube_link_head or tube_link_cross" 2>&1 | tail -2
for rep in 1 2 3; do
for o in "$@"; do
  args=""; for kv in $o; do args="$args --opt $kv"; done
  us=$(timeout 300 python bench.py --workload cc --no-cpu-baseline $args 2>/dev/null | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms_per_step']*1000,1))")
  usg=$(timeout 300 python bench.py --workload cc --graph --no-cpu-baseline $args 2>/dev/null | tail -1 | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms_per_step']*1000,1))")
  echo "$o : eager $us us   graph $usg us"
done; done | tee gpurun_out/cc_ab.txt
