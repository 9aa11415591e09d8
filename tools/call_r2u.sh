mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "attention" 2>&1 | tail -n 2
timeout -k 10 200 python tools/attn_microbench.py 32 2>&1 | grep "flash"
bash tools/collect_profiles.sh r02 > gpurun_out/collect_r02.log 2>&1
tail -n 2 gpurun_out/collect_r02.log
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/bench_r2u.log 2>&1
grep "^{" gpurun_out/bench_r2u.log | cut -c1-300
grep -o '"peak_mem_GB": [0-9.]*' gpurun_out/bench_r2u.log; grep -o '"achieved": [0-9.]*' gpurun_out/bench_r2u.log; grep -o '"flash_fwd": {[^}]*}' gpurun_out/bench_r2u.log
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/bench_r2u2.log 2>&1
grep -o '"ms_per_step": [0-9.]*' gpurun_out/bench_r2u2.log; grep -o '"achieved": [0-9.]*' gpurun_out/bench_r2u2.log; grep -o '"flash_fwd": {[^}]*}' gpurun_out/bench_r2u2.log
