#!/bin/bash
# round 3, call H: after the col_sum_g workspace fix - grouped bench, whole suite.  A GPU fault ends the call with a non-zero code.
set -o pipefail
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
step() {   # step <log> <seconds> <cmd...>
  local log=$1 secs=$2; shift 2
  timeout -k 10 $secs "$@" > gpurun_out/$log 2>&1; local rc=$?
  echo "rc=$rc" >> gpurun_out/$log
  echo "== $log rc=$rc: $(tail -3 gpurun_out/$log | cut -c1-300 | tr '\n' ' ')"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step timed out: stopping the call"; exit 1; fi
  if grep -q "Memory access fault\|GPU core dump" gpurun_out/$log; then echo "GPU fault: stopping the call"; exit 2; fi
}
step r3h_grouped_small.log 600 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "grouped or headline or device_train_b8"
CORRIF_GROUPED=1 step r3h_bench_grouped.log 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing
step r3h_suite.log 1100 python -m pytest tests -q -m gpu --durations=10
