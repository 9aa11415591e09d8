#!/bin/bash
# round 3, call N: the other BASELINE configurations on one GPU (auto schedule), bench with traffic
set -o pipefail
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
step() {   # step <log> <seconds> <cmd...>
  local log=$1 secs=$2; shift 2
  timeout -k 10 $secs "$@" > $R/gpurun_out/$log 2>&1; local rc=$?
  echo "rc=$rc" >> $R/gpurun_out/$log
  echo "== $log rc=$rc: $(tail -2 $R/gpurun_out/$log | cut -c1-700 | tr '\n' ' ')"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step timed out: stopping the call"; exit 1; fi
  if grep -q "Memory access fault\|GPU core dump" $R/gpurun_out/$log; then echo "GPU fault: stopping the call"; exit 2; fi
}
B="python bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-kernel-timing"
step r3n_cfg_b32_d8_256.log 300 $B --batch 32 --bands 8 --size 256
step r3n_cfg2_b64_d8_256.log 400 $B --batch 64 --bands 8 --size 256
step r3n_cfg4_b16_d12_512.log 400 $B --batch 16 --bands 12 --size 512
CORRIF_GROUPED=0 step r3n_cfg2_twins.log 400 $B --batch 64 --bands 8 --size 256
CORRIF_GROUPED=0 step r3n_cfg4_twins.log 400 $B --batch 16 --bands 12 --size 512
step r3n_bench_default.log 600 python bench.py
