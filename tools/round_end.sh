set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_r3d.log 2>&1
tail -3 gpurun_out/gpu_tests_r3d.log
timeout 900 python bench.py > gpurun_out/r3_final_bench.json 2> gpurun_out/r3_final_bench.err
cut -c1-300 gpurun_out/r3_final_bench.json
timeout 600 bash tools/cc_train_prof.sh r3d > gpurun_out/cc_train_prof_r3d.log 2>&1
cp gpurun_out/cctrainprof_r3d/kernel_stats.csv gpurun_out/cc_train_kernel_stats_r3d.csv
head -6 gpurun_out/cc_train_prof_r3d.log
python tools/cc_train_time.py 3 --shape 128,12,2,193,337,4
python tools/cc_train_time.py 3 --torch --shape 128,12,2,193,337,4
timeout 600 bash tools/train_prof.sh r3d > gpurun_out/train_prof_r3d.log 2>&1
cp gpurun_out/trainprof_r3d/kernel_stats.csv gpurun_out/train_kernel_stats_r3d.csv; rm -rf gpurun_out/trainprof_r3d/*/
head -3 gpurun_out/train_prof_r3d.log
timeout 600 bash tools/cfg3_prof.sh r3d > gpurun_out/cfg3_prof_r3d.log 2>&1; tail -25 gpurun_out/cfg3_prof_r3d.log | head -8
rm -rf gpurun_out/cfg3prof_r3d/*/ gpurun_out/cctrainprof_r3d/*/ 2>/dev/null; cp gpurun_out/cfg3prof_r3d/kernel_stats.csv gpurun_out/cfg3_kernel_stats_r3d.csv
