#!/bin/bash
# round 3, call D: graph-capture fix, compact-skip test, diagnostics; timeline + serial kernel stats of the grouped schedule
set -o pipefail
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
step() {   # step <log> <seconds> <cmd...>
  local log=$1 secs=$2; shift 2
  timeout -k 10 $secs "$@" > $R/gpurun_out/$log 2>&1; local rc=$?
  echo "rc=$rc" >> $R/gpurun_out/$log
  echo "== $log rc=$rc: $(tail -3 $R/gpurun_out/$log | cut -c1-400 | tr '\n' ' ')"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step timed out: stopping the call"; exit 1; fi
}
step r3d_graph_model.log 300 python -X faulthandler tools/probe/graph_fork2.py model
step r3d_tests.log 1100 python -m pytest tests/test_model_gpu.py tests/test_train_gpu.py -x -q -m gpu -k "compact_skip or hip_graph or determinism" --durations=8
step r3d_local_error.log 600 python tools/local_error.py
O=$R/gpurun_out/prof_r3d
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing"
step r3d_prof_conc.log 400 rocprofv3 --kernel-trace --stats -d $O/concurrent -o c --output-format csv -- python $R/bench.py $ARGS
python $R/tools/timeline.py $O/concurrent/c_kernel_trace.csv 255 > $O/timeline_grouped.txt 2>&1
export CORRIF_SERIAL=1
step r3d_prof_serial.log 400 rocprofv3 --kernel-trace --stats -d $O/serial -o s --output-format csv -- python $R/bench.py $ARGS
unset CORRIF_SERIAL
export CORRIF_GROUPED=0
step r3d_prof_conc_twins.log 400 rocprofv3 --kernel-trace --stats -d $O/concurrent_twins -o c --output-format csv -- python $R/bench.py $ARGS
python $R/tools/timeline.py $O/concurrent_twins/c_kernel_trace.csv 245 > $O/timeline_twins.txt 2>&1
rm -f $O/*/*_kernel_trace.csv
ls $O $O/serial | head
