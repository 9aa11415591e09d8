#!/bin/bash
# split-bf16: per-shape timings both ways
set -o pipefail
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
step() { local log=$1 secs=$2; shift 2; timeout -k 10 $secs "$@" > gpurun_out/$log 2>&1; local rc=$?; echo "rc=$rc" >> gpurun_out/$log; echo "== $log rc=$rc: $(tail -3 gpurun_out/$log | cut -c1-300 | tr '\n' ' ')"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi; if grep -q "Memory access fault\|GPU core dump" gpurun_out/$log; then exit 2; fi; }
step r3t_split.log 400 python bench.py --steps 4 --warmup 3 --no-cpu-baseline --dump-shapes gpurun_out/shapes_r3t_split.jsonl
CORRIF_SPLIT_BF16=0 step r3t_f32.log 400 python bench.py --steps 4 --warmup 3 --no-cpu-baseline --dump-shapes gpurun_out/shapes_r3t_f32.jsonl
