mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "attention" > gpurun_out/t_r2f.log 2>&1
echo "pytest rc=$?" >> gpurun_out/t_r2f.log
tail -n 15 gpurun_out/t_r2f.log | cut -c1-300
timeout -k 10 300 python tools/attn_microbench.py 32 > gpurun_out/attn_mb.log 2>&1
echo "rc=$?" >> gpurun_out/attn_mb.log
cat gpurun_out/attn_mb.log
