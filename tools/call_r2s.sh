mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "backward_statistics or batch_norm or conv3d or gemm" > gpurun_out/t_r2s.log 2>&1
echo "pytest rc=$?" >> gpurun_out/t_r2s.log
tail -n 12 gpurun_out/t_r2s.log | cut -c1-400
timeout -k 10 900 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "multi_consumer or determinism or tame_train or full_gradient or kaiming" >> gpurun_out/t_r2s.log 2>&1
tail -n 6 gpurun_out/t_r2s.log | cut -c1-600
for v in 1 0 1 0; do
CORRIF_BWD_STATS=$v timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing > gpurun_out/bst_$v.log 2>&1
echo "bwd_stats=$v: $(grep '^{' gpurun_out/bst_$v.log | cut -c60-150)"
done
