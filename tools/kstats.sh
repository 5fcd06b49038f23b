#!/bin/bash
# per-kernel average durations of the bench workload over the TIMED region only:
#   tools/kstats.sh <tag> [bench args...]  -> gpurun_out/kstats_<tag>/{bench.log, kernel_stats_timed.csv, kernel_stats.csv (rocprofv3's own)}
# bench.py runs with --no-extras --no-qkav --no-stages: no truncated (spatial_only) launches, and the last K launches of every layer
# kernel in the raw trace are exactly the K timed steps (tools/trace_reduce.py); the line's launch_us comes from the same run.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=$1; shift
OUT=$R/gpurun_out/kstats_$tag
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/run -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --no-qkav --no-stages "$@" > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-260
python3 $R/tools/trace_reduce.py $OUT/run $OUT/bench.log --out $OUT/kernel_stats_timed.csv
cp $OUT/run/*/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
# keep the merged-back scratch small: the raw trace stays on the box unless KEEP_TRACE=1
if [ -z "$KEEP_TRACE" ]; then rm -rf $OUT/run; fi
