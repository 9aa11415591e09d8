#!/bin/bash
# round 3, call G: patch-epilogue statistics + fused depth-class add, grouped == twins bit-identity, whole suite, bench A/B
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
step r3g_quick.log 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "patch_conv_epilogue or k_split or stream_k or grouped or full_gradient or compact_skip or conv3d"
step r3g_bench.log 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline
CORRIF_GROUPED=1 step r3g_bench_grouped.log 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing
step r3g_suite.log 1100 python -m pytest tests -q -m gpu --durations=10
