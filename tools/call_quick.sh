mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_model_gpu.py tests/test_train_gpu.py -m gpu -q -x -k "determinism or tame_train_b2 or tame_eval_b3_d3_64 or multi_consumer or stage_taps or hip_graph or module_surface" 2>&1 | tail -n 3
for v in 1 0 1 0; do
CORRIF_INTERLEAVE=$v timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing > gpurun_out/il_$v.log 2>&1
echo "interleave=$v: $(grep '^{' gpurun_out/il_$v.log | cut -c60-150)"
done
