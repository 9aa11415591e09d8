mkdir -p gpurun_out
timeout -k 10 700 python tools/grad_diag.py 4 4 224 1.0 4 > gpurun_out/diag_b4.log 2>&1
echo "rc=$?" >> gpurun_out/diag_b4.log
timeout -k 10 200 python tools/grad_diag.py 2 3 64 1.0 0 > gpurun_out/diag_b2.log 2>&1
echo "rc=$?" >> gpurun_out/diag_b2.log
CORRIF_STREAM_K=0 timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --dump-shapes gpurun_out/shapes_r2d_nosk.jsonl > gpurun_out/bench_r2d_nosk.log 2>&1
echo "rc=$?" >> gpurun_out/bench_r2d_nosk.log
tail -n 1 gpurun_out/bench_r2d_nosk.log | cut -c1-300
grep "==" gpurun_out/diag_b4.log gpurun_out/diag_b2.log
