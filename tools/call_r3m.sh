#!/bin/bash
# round 3, call M: exact-step timeline of the default schedule; which host ops issue the D2D copies
set -o pipefail
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
step() {   # step <log> <seconds> <cmd...>
  local log=$1 secs=$2; shift 2
  timeout -k 10 $secs "$@" > $R/gpurun_out/$log 2>&1; local rc=$?
  echo "rc=$rc" >> $R/gpurun_out/$log
  echo "== $log rc=$rc: $(tail -4 $R/gpurun_out/$log | cut -c1-400 | tr '\n' ' ')"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step timed out: stopping the call"; exit 1; fi
  if grep -q "Memory access fault\|GPU core dump" $R/gpurun_out/$log; then echo "GPU fault: stopping the call"; exit 2; fi
}
step r3m_copy_census.log 400 python tools/copy_census.py 8
O=$R/gpurun_out/prof_r3m
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
step r3m_prof_conc.log 400 rocprofv3 --kernel-trace --stats -d $O/concurrent -o c --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing
python $R/tools/timeline.py $O/concurrent/c_kernel_trace.csv 244 > $O/timeline.txt 2>&1
rm -f $O/*/*_kernel_trace.csv
head -8 $O/timeline.txt
