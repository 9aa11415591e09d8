#!/bin/bash
# round 3, call C: grouped encoder + side-stream weight gradients + stream-K (A/B), graph-capture fix, any-ratio compact skip, diagnostics
set -o pipefail
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
step() {   # step <log> <seconds> <cmd...>
  local log=$1 secs=$2; shift 2
  timeout -k 10 $secs "$@" > gpurun_out/$log 2>&1; local rc=$?
  echo "rc=$rc" >> gpurun_out/$log
  echo "== $log rc=$rc: $(tail -3 gpurun_out/$log | cut -c1-400 | tr '\n' ' ')"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step timed out: stopping the call"; exit 1; fi
}
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing"
step r3c_bench_default.log 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline
CORRIF_SIDE_WGRAD_G=0 step r3c_bench_noside.log 300 $B
CORRIF_STREAM_K_G=0 step r3c_bench_nosk.log 300 $B
CORRIF_GROUPED=0 step r3c_bench_twins.log 300 $B
step r3c_graph_origin_lane.log 200 python -X faulthandler tools/probe/graph_fork2.py origin_lane
step r3c_graph_model.log 300 python -X faulthandler tools/probe/graph_fork2.py model
step r3c_tests.log 1100 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_train_gpu.py -x -q -m gpu -k "depth_class or compact_skip or hip_graph or determinism or grouped or stage_taps or tame_train_b2_d3_64" --durations=8
step r3c_local_error.log 600 python tools/local_error.py
