mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "conv3d" > gpurun_out/t_r2l.log 2>&1
echo "pytest rc=$?" >> gpurun_out/t_r2l.log
tail -n 5 gpurun_out/t_r2l.log | cut -c1-400
timeout -k 10 600 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "compact or tame_train_b1_d4" >> gpurun_out/t_r2l.log 2>&1
tail -n 3 gpurun_out/t_r2l.log | cut -c1-300
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --dump-shapes gpurun_out/shapes_r2l.jsonl > gpurun_out/bench_r2l.log 2>&1
echo "rc=$?" >> gpurun_out/bench_r2l.log
grep "^{" gpurun_out/bench_r2l.log | cut -c1-420
