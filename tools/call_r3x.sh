#!/bin/bash
# full GPU suite, smoke, bench (with the issued-work roofline block), product GEMM vs lab shapes
set -o pipefail
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
step() { local log=$1 secs=$2; shift 2; timeout -k 10 $secs "$@" > gpurun_out/$log 2>&1; local rc=$?; echo "rc=$rc" >> gpurun_out/$log; echo "== $log rc=$rc: $(tail -3 gpurun_out/$log | cut -c1-330 | tr '\n' ' ')"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi; if grep -q "Memory access fault\|GPU core dump" gpurun_out/$log; then exit 2; fi; }
step r3x_suite.log 900 python -m pytest tests -q -m gpu --durations=5
step r3x_smoke.log 300 python -c "import __graft_entry__ as g; g.smoke()"
step r3x_bench.log 600 python bench.py --steps 10 --warmup 4 --no-cpu-baseline
step r3x_gemm_vs_lab.log 300 python tools/gemm_vs_lab.py
step r3x_mm2.log 300 python tools/mm2_norm_probe.py
