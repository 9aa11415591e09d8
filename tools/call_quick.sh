mkdir -p gpurun_out
timeout -k 10 300 python bench.py --no-cpu-baseline --no-kernel-timing > gpurun_out/bq1.log 2>&1
grep -o '"ms_per_step": [0-9.]*' gpurun_out/bq1.log | head -1; grep -o '"peak_mem_GB": [0-9.]*' gpurun_out/bq1.log
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/bq2.log 2>&1
grep -o '"ms_per_step": [0-9.]*' gpurun_out/bq2.log | head -1; grep -o '"peak_mem_GB": [0-9.]*' gpurun_out/bq2.log
timeout -k 10 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 6 > gpurun_out/bq3.log 2>&1
grep -o '"ms_per_step": [0-9.]*' gpurun_out/bq3.log | head -1; grep -o '"peak_mem_GB": [0-9.]*' gpurun_out/bq3.log
