mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_lab.hip -o gpurun_out/gemm_lab > gpurun_out/lab_build.log 2>&1 && timeout -k 10 300 gpurun_out/gemm_lab > gpurun_out/gemm_lab.log 2>&1
echo "lab rc=$?" >> gpurun_out/gemm_lab.log
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q > gpurun_out/t_r2b_kernels.log 2>&1
echo "rc=$?" >> gpurun_out/t_r2b_kernels.log
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --dump-shapes gpurun_out/shapes_r2b_sk.jsonl > gpurun_out/bench_r2b_sk.log 2>&1
echo "rc=$?" >> gpurun_out/bench_r2b_sk.log
CORRIF_STREAM_K=0 timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --dump-shapes gpurun_out/shapes_r2b_nosk.jsonl > gpurun_out/bench_r2b_nosk.log 2>&1
echo "rc=$?" >> gpurun_out/bench_r2b_nosk.log
tail -n 3 gpurun_out/t_r2b_kernels.log
tail -n 2 gpurun_out/bench_r2b_sk.log | cut -c1-400
tail -n 2 gpurun_out/bench_r2b_nosk.log | cut -c1-400
