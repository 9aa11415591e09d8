#!/bin/bash
# round 3, call J: whole GPU suite on the final code, then the round's profile set (kernel stats concurrent / serial, timeline, HBM traffic, MFMA busy)
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
step r3j_suite.log 900 python -m pytest tests -q -m gpu --durations=10
step r3j_collect.log 1100 bash tools/collect_profiles.sh r03
CORRIF_GROUPED=1 CORRIF_SERIAL=1 step r3j_grouped_serial.log 300 bash -c "cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r03/serial_grouped -o s --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing; rm -f $R/gpurun_out/prof_r03/serial_grouped/*_kernel_trace.csv"
