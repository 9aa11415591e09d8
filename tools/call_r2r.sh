mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_model_gpu.py tests/test_train_gpu.py -m gpu -q -x -k "determinism or tame_train_b2 or multi_consumer or train_steps or hip_graph or compact" > gpurun_out/t_r2r.log 2>&1
echo "pytest rc=$?" >> gpurun_out/t_r2r.log
tail -n 4 gpurun_out/t_r2r.log | cut -c1-300
for v in 1 0 1 0; do
CORRIF_SIDE_WGRAD=$v timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing > gpurun_out/side_$v.log 2>&1
echo "side=$v: $(grep '^{' gpurun_out/side_$v.log | cut -c60-150) $(grep -o '"peak_mem_GB": [0-9.]*' gpurun_out/side_$v.log)"
done
