mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "attention" > gpurun_out/t_r2g.log 2>&1
echo "pytest rc=$?" >> gpurun_out/t_r2g.log
tail -n 5 gpurun_out/t_r2g.log | cut -c1-300
timeout -k 10 300 python tools/attn_microbench.py 32 > gpurun_out/attn_mb.log 2>&1
echo "rc=$?" >> gpurun_out/attn_mb.log
grep -v amdgpu gpurun_out/attn_mb.log
timeout -k 10 600 python -m pytest tests/test_model_gpu.py tests/test_train_gpu.py -m gpu -q -x -k "tame or determinism or dropout or train_steps or hip_graph" > gpurun_out/t_r2g_model.log 2>&1
echo "pytest rc=$?" >> gpurun_out/t_r2g_model.log
tail -n 5 gpurun_out/t_r2g_model.log | cut -c1-300
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --dump-shapes gpurun_out/shapes_r2g.jsonl > gpurun_out/bench_r2g.log 2>&1
echo "rc=$?" >> gpurun_out/bench_r2g.log
grep "^{" gpurun_out/bench_r2g.log | cut -c1-700
