mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/t_suite.log 2>&1
echo "pytest rc=$?" >> gpurun_out/t_suite.log
tail -n 12 gpurun_out/t_suite.log | cut -c1-300
