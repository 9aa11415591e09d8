#!/bin/bash
# split-bf16 (pair split, unmasked interior staging): kernel parity, bench, stream-K A/B
set -o pipefail
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
step() { local log=$1 secs=$2; shift 2; timeout -k 10 $secs "$@" > gpurun_out/$log 2>&1; local rc=$?; echo "rc=$rc" >> gpurun_out/$log; echo "== $log rc=$rc: $(tail -3 gpurun_out/$log | cut -c1-330 | tr '\n' ' ')"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi; if grep -q "Memory access fault\|GPU core dump" gpurun_out/$log; then exit 2; fi; }
step r3w_kernels.log 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu
step r3w_bench.log 400 python bench.py --steps 10 --warmup 4 --no-cpu-baseline
CORRIF_STREAM_K_LONG=1 step r3w_bench_sklong.log 400 python bench.py --steps 10 --warmup 4 --no-cpu-baseline
CORRIF_STREAM_K=1 step r3w_bench_sk.log 400 python bench.py --steps 10 --warmup 4 --no-cpu-baseline
CORRIF_GROUPED=1 step r3w_bench_grouped.log 400 python bench.py --steps 10 --warmup 4 --no-cpu-baseline
