mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "stem" > gpurun_out/t_r2n.log 2>&1
echo "pytest rc=$?" >> gpurun_out/t_r2n.log
tail -n 12 gpurun_out/t_r2n.log | cut -c1-300
timeout -k 10 600 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "tame_train_b1_d4 or tame_train_b2 or determinism" >> gpurun_out/t_r2n.log 2>&1
tail -n 3 gpurun_out/t_r2n.log | cut -c1-300
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --dump-shapes gpurun_out/shapes_r2n.jsonl > gpurun_out/bench_r2n.log 2>&1
echo "rc=$?" >> gpurun_out/bench_r2n.log
grep "^{" gpurun_out/bench_r2n.log | cut -c1-420
