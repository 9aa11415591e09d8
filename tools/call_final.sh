mkdir -p gpurun_out
timeout -k 10 500 python bench.py > gpurun_out/bench_final.log 2>&1
echo "bench rc=$?" >> gpurun_out/bench_final.log
grep "^{" gpurun_out/bench_final.log | cut -c1-600
bash tools/collect_profiles.sh r02 > gpurun_out/collect_r02.log 2>&1
tail -n 3 gpurun_out/collect_r02.log
timeout -k 10 1100 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/t_final.log 2>&1
echo "pytest rc=$?" >> gpurun_out/t_final.log
tail -n 12 gpurun_out/t_final.log | cut -c1-300
