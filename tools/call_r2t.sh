mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "headline or full_gradient" > gpurun_out/t_r2t.log 2>&1
echo "pytest rc=$?" >> gpurun_out/t_r2t.log
tail -n 8 gpurun_out/t_r2t.log | cut -c1-500
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/bench_r2t.log 2>&1
grep "^{" gpurun_out/bench_r2t.log | cut -c1-800
