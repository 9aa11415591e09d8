mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_mmvit2_gpu.py -m gpu -q -x -k "depth_class or compact or tame or determinism or multi_consumer or mmvit2 or reference_fixture" > gpurun_out/t_r2k.log 2>&1
echo "pytest rc=$?" >> gpurun_out/t_r2k.log
tail -n 8 gpurun_out/t_r2k.log | cut -c1-600
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --dump-shapes gpurun_out/shapes_r2k.jsonl > gpurun_out/bench_r2k.log 2>&1
echo "rc=$?" >> gpurun_out/bench_r2k.log
grep "^{" gpurun_out/bench_r2k.log | cut -c1-900
