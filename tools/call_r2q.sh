mkdir -p gpurun_out
for v in 2 3 4 2 3; do
CORRIF_DECODER_SPLIT=$v timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing > gpurun_out/split_$v.log 2>&1
echo "split=$v: $(grep '^{' gpurun_out/split_$v.log | cut -c60-150)"
done
