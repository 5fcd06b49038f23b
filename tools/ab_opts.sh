#!/bin/bash
# tools/ab_opts.sh rounds "optsA" "optsB" ... -- wall time per step of bench.py (1000 steps) for several library option sets, interleaved
# `rounds` times on this box (timings of different boxes are not comparable).  An option set is a string of bench.py flags, e.g.
# "--opt no_vrow=1"; "" is the default configuration.
R=${GRAFT_REPO_ROOT:-$(dirname $(dirname $(readlink -f $0)))}
ROUNDS=$1; shift
for r in $(seq 1 $ROUNDS); do
  for o in "$@"; do
    python3 $R/bench.py --no-extras --no-cpu-baseline $o 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('%-40s %8.2f us/step  %s' % ('[$o]', j['ms_per_step'] * 1e3, {k: round(v, 1) for k, v in j['roofline']['stage_us'].items()}))"
  done
done
