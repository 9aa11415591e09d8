#!/bin/bash
# round-end check: the GPU suite, smoke(), the default bench line
set -o pipefail
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
step() { local log=$1 secs=$2; shift 2; timeout -k 10 $secs "$@" > gpurun_out/$log 2>&1; local rc=$?; echo "rc=$rc" >> gpurun_out/$log; echo "== $log rc=$rc: $(tail -3 gpurun_out/$log | cut -c1-600 | tr '\n' ' ')"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi; if grep -q "Memory access fault\|GPU core dump" gpurun_out/$log; then exit 2; fi; }
step final_suite.log 1000 python -m pytest tests -q -m gpu --durations=8
step final_smoke.log 300 python -c "import __graft_entry__ as g; g.smoke()"
step final_bench.log 900 python bench.py
