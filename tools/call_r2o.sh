mkdir -p gpurun_out
for v in 1 0 1 0; do
CORRIF_STREAM_K_LONG=$v timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing > gpurun_out/skl_$v.log 2>&1
echo "skl=$v: $(grep '^{' gpurun_out/skl_$v.log | cut -c60-150)"
done
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "stream_k or conv3d" 2>&1 | tail -n 2
