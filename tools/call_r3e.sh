#!/bin/bash
# round 3, call E: two-level K accumulation (accuracy + speed A/B), grouped vs twins across batch sizes
set -o pipefail
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
step() {   # step <log> <seconds> <cmd...>
  local log=$1 secs=$2; shift 2
  timeout -k 10 $secs "$@" > gpurun_out/$log 2>&1; local rc=$?
  echo "rc=$rc" >> gpurun_out/$log
  echo "== $log rc=$rc: $(tail -3 gpurun_out/$log | cut -c1-300 | tr '\n' ' ')"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step timed out: stopping the call"; exit 1; fi
}
step r3e_group.log 300 python tools/group_microbench.py
step r3e_local_error.log 600 python tools/local_error.py
step r3e_grad_diag.log 900 python tools/grad_diag.py 2 3 64 1.0 11
step r3e_kernels.log 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "conv3d or grouped or stream_k or linear or epilogue"
step r3e_bench_full.log 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing"
for b in 32 16 8 4 2; do
  step r3e_bench_g_b$b.log 300 $B --batch $b
  CORRIF_GROUPED=0 step r3e_bench_t_b$b.log 300 $B --batch $b
done
