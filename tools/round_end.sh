set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 bash tools/timeline.sh > gpurun_out/timeline.log 2>&1
timeout 600 bash tools/train_prof.sh r3c > gpurun_out/train_prof_r3c.log 2>&1
cp gpurun_out/trainprof_r3c/kernel_stats.csv gpurun_out/train_kernel_stats_r3c.csv; rm -rf gpurun_out/trainprof_r3c/*/
timeout 900 python bench.py > gpurun_out/r3_final_bench.json 2> gpurun_out/r3_final_bench.err
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_r3c.log 2>&1
tail -5 gpurun_out/gpu_tests_r3c.log
tail -12 gpurun_out/train_prof_r3c.log
cut -c1-400 gpurun_out/r3_final_bench.json
tail -30 gpurun_out/timeline.log
