#!/bin/bash
# round 3, call K: the profile set of the PER-MODALITY schedule (what the B=32 timed region runs); copy census; eval throughput
set -o pipefail
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
step() {   # step <log> <seconds> <cmd...>
  local log=$1 secs=$2; shift 2
  timeout -k 10 $secs "$@" > $R/gpurun_out/$log 2>&1; local rc=$?
  echo "rc=$rc" >> $R/gpurun_out/$log
  echo "== $log rc=$rc: $(tail -3 $R/gpurun_out/$log | cut -c1-300 | tr '\n' ' ')"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step timed out: stopping the call"; exit 1; fi
  if grep -q "Memory access fault\|GPU core dump" $R/gpurun_out/$log; then echo "GPU fault: stopping the call"; exit 2; fi
}
export CORRIF_GROUPED=0
step r3k_collect.log 1100 bash tools/collect_profiles.sh r03
unset CORRIF_GROUPED
step r3k_copy_census.log 300 python tools/copy_census.py 4
step r3k_eval.log 400 python tools/eval_throughput.py
