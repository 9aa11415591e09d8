#!/bin/bash
# split-bf16: lab ablations, kernel parity, bench
set -o pipefail
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
step() { local log=$1 secs=$2; shift 2; timeout -k 10 $secs "$@" > gpurun_out/$log 2>&1; local rc=$?; echo "rc=$rc" >> gpurun_out/$log; echo "== $log rc=$rc: $(tail -3 gpurun_out/$log | cut -c1-400 | tr '\n' ' ')"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi; if grep -q "Memory access fault\|GPU core dump" gpurun_out/$log; then exit 2; fi; }
step split_lab2.txt 240 tools/probe/split_lab.bin
step r3u_kernels.log 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu
step r3u_bench_split.log 400 python bench.py --steps 10 --warmup 4 --no-cpu-baseline
