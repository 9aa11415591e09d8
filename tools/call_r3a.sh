#!/bin/bash
# round 3, call A: new parity tests + diagnostics + grouped-launch microbenchmark + the 12-band CPU fixture.
# A step that times out (or is killed) ends the call: no further GPU step is started behind a hung one.
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
step r3a_intercorr.log 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k inter_corr
step r3a_headline.log 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "headline_batch or device_train_b8" --durations=5
step r3a_group.log 300 python tools/group_microbench.py
step r3a_local_error.log 600 python tools/local_error.py
CORRIF_GOLDEN_DTYPES=f32 step r3a_golden.log 1000 python tests/golden/make_golden_large.py gpurun_out/golden oracle_train_b2_d12_512
step r3a_bench.log 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dump-shapes gpurun_out/shapes_r3a.jsonl
step r3a_graph_raw.log 200 python -X faulthandler tools/probe/graph_fork2.py raw
step r3a_graph_alloc.log 200 python -X faulthandler tools/probe/graph_fork2.py alloc
step r3a_graph_model.log 300 python -X faulthandler tools/probe/graph_fork2.py model
