mkdir -p gpurun_out
run() { # name, env..., args
  n=$1; shift
  env "$@" > /dev/null 2>&1
}
CORRIF_AUTO_STREAMS=1 timeout -k 10 400 python bench.py --batch 64 --bands 8 --size 256 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing > gpurun_out/cfg2_auto.log 2>&1
grep "^{" gpurun_out/cfg2_auto.log | cut -c60-160; grep -o '"peak_mem_GB": [0-9.]*' gpurun_out/cfg2_auto.log
CORRIF_AUTO_STREAMS=0 timeout -k 10 400 python bench.py --batch 64 --bands 8 --size 256 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing > gpurun_out/cfg2_multi.log 2>&1
grep "^{" gpurun_out/cfg2_multi.log | cut -c60-160; grep -o '"peak_mem_GB": [0-9.]*' gpurun_out/cfg2_multi.log
CORRIF_AUTO_STREAMS=1 timeout -k 10 400 python bench.py --batch 16 --bands 12 --size 512 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing > gpurun_out/cfg4_auto.log 2>&1
grep "^{" gpurun_out/cfg4_auto.log | cut -c60-160; grep -o '"peak_mem_GB": [0-9.]*' gpurun_out/cfg4_auto.log
CORRIF_AUTO_STREAMS=0 timeout -k 10 400 python bench.py --batch 16 --bands 12 --size 512 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing > gpurun_out/cfg4_multi.log 2>&1
grep "^{" gpurun_out/cfg4_multi.log | cut -c60-160; grep -o '"peak_mem_GB": [0-9.]*' gpurun_out/cfg4_multi.log
