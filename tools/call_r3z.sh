#!/bin/bash
# the other BASELINE configurations with the split-bf16 GEMM family (per-GPU shards), and small-batch eval throughput
set -o pipefail
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
step() { local log=$1 secs=$2; shift 2; timeout -k 10 $secs "$@" > gpurun_out/$log 2>&1; local rc=$?; echo "rc=$rc" >> gpurun_out/$log; echo "== $log rc=$rc: $(tail -2 gpurun_out/$log | cut -c1-500 | tr '\n' ' ')"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi; if grep -q "Memory access fault\|GPU core dump" gpurun_out/$log; then exit 2; fi; }
B="python bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-kernel-timing"
step r3z_cfg2_b64_d8_256.log 400 $B --batch 64 --bands 8 --size 256
step r3z_cfg4_b16_d12_512.log 400 $B --batch 16 --bands 12 --size 512
step r3z_cfg_b32_d8_256.log 400 $B --batch 32 --bands 8 --size 256
step r3z_eval.log 400 python tools/eval_throughput.py
