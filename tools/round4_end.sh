#!/bin/bash
# round-4 evidence set: bench line, kernel stats of the bench command, HBM traffic (PMC), cfg 3 kernel stats -> gpurun_out/r4e_*
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; mkdir -p gpurun_out
timeout 900 python3 bench.py > gpurun_out/r4e_bench.json 2> gpurun_out/r4e_bench.err; cut -c1-400 gpurun_out/r4e_bench.json
timeout 600 bash tools/kstats.sh r4e --no-extras > gpurun_out/r4e_kstats.log 2>&1; tail -6 gpurun_out/r4e_kstats.log
cp gpurun_out/kstats_r4e/kernel_stats.csv gpurun_out/r4e_kernel_stats.csv; rm -rf gpurun_out/kstats_r4e/*/
timeout 600 bash tools/pmc_traffic.sh > gpurun_out/r4e_pmc.log 2>&1; tail -5 gpurun_out/r4e_pmc.log
cp gpurun_out/pmc_traffic/traffic.json gpurun_out/r4e_traffic.json; rm -rf gpurun_out/pmc_traffic/*/
timeout 600 bash tools/cfg3_prof.sh r4e > gpurun_out/r4e_cfg3.log 2>&1; tail -28 gpurun_out/r4e_cfg3.log | head -30
cp gpurun_out/cfg3prof_r4e/kernel_stats.csv gpurun_out/r4e_cfg3_kernel_stats.csv; rm -rf gpurun_out/cfg3prof_r4e/*/
