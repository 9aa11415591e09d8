#!/bin/bash
# round 3, call I: grouped == per-modality bit identity; the N > 1 gradient path with one rank (overhead vs dp1); default bench; ATen launch census
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
step r3i_tests.log 900 python -m pytest tests/test_model_gpu.py tests/test_kernels_gpu.py tests/test_train_gpu.py -x -q -m gpu -k "grouped or trilinear or layer_norm or reducer or rehearsal"
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing"
step r3i_bench_dp1.log 300 $B
CORRIF_FORCE_COLLECTIVE=1 step r3i_bench_forced.log 300 $B
step r3i_bench_dp1_b.log 300 $B
CORRIF_FORCE_COLLECTIVE=1 step r3i_bench_forced_b.log 300 $B
step r3i_bench_full.log 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline
O=$R/gpurun_out/prof_r3i
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export CORRIF_FORCE_COLLECTIVE=1
step r3i_prof_forced.log 400 rocprofv3 --kernel-trace --stats -d $O/forced -o f --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing
rm -f $O/*/*_kernel_trace.csv
