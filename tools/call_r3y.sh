#!/bin/bash
# per-layer local error with the split loop on / off; the two model tests that failed
set -o pipefail
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
step() { local log=$1 secs=$2; shift 2; timeout -k 10 $secs "$@" > gpurun_out/$log 2>&1; local rc=$?; echo "rc=$rc" >> gpurun_out/$log; echo "== $log rc=$rc: $(tail -3 gpurun_out/$log | cut -c1-330 | tr '\n' ' ')"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi; if grep -q "Memory access fault\|GPU core dump" gpurun_out/$log; then exit 2; fi; }
step r3y_local_split.log 500 python tools/local_error.py 2 3 64 11
CORRIF_SPLIT_BF16=0 step r3y_local_f32.log 500 python tools/local_error.py 2 3 64 11
step r3y_kernels.log 500 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x
step r3y_tests.log 600 python -m pytest tests/test_model_gpu.py -q -m gpu -k "grouped_encoders or full_gradient"
