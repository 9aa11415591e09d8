mkdir -p gpurun_out
timeout -k 10 300 python tools/aten_ops.py 8 > gpurun_out/aten_ops.log 2>&1
echo "rc=$?" >> gpurun_out/aten_ops.log
NO_REDUCER=1 timeout -k 10 300 python tools/aten_ops.py 8 > gpurun_out/aten_ops_noreducer.log 2>&1
echo "rc=$?" >> gpurun_out/aten_ops_noreducer.log
timeout -k 10 1100 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "kaiming or full_gradient or device_oracle or large_baseline" > gpurun_out/t_r2e.log 2>&1
echo "pytest rc=$?" >> gpurun_out/t_r2e.log
tail -n 8 gpurun_out/t_r2e.log
tail -n 5 gpurun_out/aten_ops.log
