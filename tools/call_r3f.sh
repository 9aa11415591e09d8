#!/bin/bash
# round 3, call F: whole GPU suite; local error with the two-level accumulation; default bench (auto encoder schedule, HBM families)
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
step r3f_bench.log 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline
CORRIF_KSPLIT=1 step r3f_local_error_ks.log 600 python tools/local_error.py
step r3f_suite.log 1150 python -m pytest tests -q -m gpu --durations=12
