#!/bin/bash
# separable trilinear adjoint: kernel tests, A/B bench, kernel stats
set -o pipefail
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
step() { local log=$1 secs=$2; shift 2; timeout -k 10 $secs "$@" > gpurun_out/$log 2>&1; local rc=$?; echo "rc=$rc" >> gpurun_out/$log; echo "== $log rc=$rc: $(tail -3 gpurun_out/$log | cut -c1-900 | tr '\n' ' ')"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi; if grep -q "Memory access fault\|GPU core dump" gpurun_out/$log; then exit 2; fi; }
step r3m_tests.log 400 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "trilinear or nearest"
step r3m_bench_sep.log 400 python bench.py --steps 10 --warmup 4 --no-cpu-baseline
CORRIF_TRILINEAR_SEP=0 step r3m_bench_gather.log 400 python bench.py --steps 10 --warmup 4 --no-cpu-baseline
step r3m_bench_sep2.log 400 python bench.py --steps 10 --warmup 4 --no-cpu-baseline
cd /tmp && export TMPDIR=/tmp
step r3m_prof.log 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r3m_prof -o sep -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timing
