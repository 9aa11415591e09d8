#!/bin/bash
# round 3, call B: grouped encoder launches - kernel tests, model parity, microbenchmark, A/B bench; diagnostics; graph probes
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
step r3b_kernels.log 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "grouped or inter_corr or batch_norm or conv3d"
step r3b_model_quick.log 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "tame_train_b2_d3_64 or determinism or multi_consumer or full_gradient"
step r3b_group.log 300 python tools/group_microbench.py
step r3b_bench_grouped.log 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dump-shapes gpurun_out/shapes_r3b.jsonl
CORRIF_GROUPED=0 step r3b_bench_twins.log 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dump-shapes gpurun_out/shapes_r3b_twins.jsonl
step r3b_local_error.log 600 python tools/local_error.py
step r3b_headline.log 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "headline_batch or device_train_b8 or d12_512" --durations=5
for c in origin join1 onelane; do
  step r3b_graph_$c.log 200 python -X faulthandler tools/probe/graph_fork2.py $c
done
