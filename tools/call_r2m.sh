mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/t_r2m.log 2>&1
echo "pytest rc=$?" >> gpurun_out/t_r2m.log
tail -n 16 gpurun_out/t_r2m.log | cut -c1-300
