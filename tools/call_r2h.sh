mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_model_gpu.py tests/test_kernels_gpu.py -m gpu -q -x -k "multi_consumer or determinism or tame_train_b2 or conv3d or gemm or linear or module_surface" > gpurun_out/t_r2h.log 2>&1
echo "pytest rc=$?" >> gpurun_out/t_r2h.log
tail -n 6 gpurun_out/t_r2h.log | cut -c1-400
for tap in 1 0 1 0; do
CORRIF_GRAD_TAP=$tap timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing > gpurun_out/bench_r2h_tap$tap.log 2>&1
grep "^{" gpurun_out/bench_r2h_tap$tap.log | cut -c60-140
done
