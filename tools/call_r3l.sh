#!/bin/bash
# round 3, call L: the TORCH_LIBRARY binding - equality with the ctypes binding, eval fixtures, HIP graph, eval throughput A/B
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
step r3l_tests.log 900 python -m pytest tests/test_train_gpu.py tests/test_model_gpu.py tests/test_mmvit2_gpu.py -x -q -m gpu -k "torch_library or hip_graph or eval or other_baseline or stage_taps or evaluate or per_image or mm2"
step r3l_eval.log 600 python tools/eval_throughput.py
