mkdir -p gpurun_out
bash tools/collect_profiles.sh r02 > gpurun_out/collect_r02.log 2>&1
tail -n 12 gpurun_out/collect_r02.log
timeout -k 10 300 python bench.py --batch 64 --bands 8 --size 256 --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing > gpurun_out/cfg2.log 2>&1
echo "cfg2: $(grep '^{' gpurun_out/cfg2.log | cut -c60-150) $(grep -o '"peak_mem_GB": [0-9.]*' gpurun_out/cfg2.log)"
timeout -k 10 300 python bench.py --batch 16 --bands 12 --size 512 --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing > gpurun_out/cfg4.log 2>&1
echo "cfg4: $(grep '^{' gpurun_out/cfg4.log | cut -c60-150) $(grep -o '"peak_mem_GB": [0-9.]*' gpurun_out/cfg4.log)"
timeout -k 10 300 python bench.py --batch 32 --bands 8 --size 256 --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing > gpurun_out/cfg2h.log 2>&1
echo "B32 8x256: $(grep '^{' gpurun_out/cfg2h.log | cut -c60-150) $(grep -o '"peak_mem_GB": [0-9.]*' gpurun_out/cfg2h.log)"
timeout -k 10 200 python tools/eval_throughput.py > gpurun_out/eval_tp.log 2>&1; grep -v amdgpu gpurun_out/eval_tp.log | tail -4
