# round-2 validation call: full GPU suite, bench with per-shape timings, serial + concurrent kernel stats
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -m gpu -q --durations=15 > gpurun_out/t_r2c.log 2>&1
echo "pytest rc=$?" >> gpurun_out/t_r2c.log
tail -n 25 gpurun_out/t_r2c.log
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --dump-shapes gpurun_out/shapes_r2c.jsonl > gpurun_out/bench_r2c.log 2>&1
echo "bench rc=$?" >> gpurun_out/bench_r2c.log
tail -n 2 gpurun_out/bench_r2c.log | cut -c1-1500
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_r2c
mkdir -p $O
CORRIF_SERIAL=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/serial -o s --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $O/serial.log 2>&1 || echo "serial prof failed"
rm -f $O/serial/*_kernel_trace.csv
ls $O/serial
