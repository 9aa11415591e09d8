#!/bin/bash
# Matrix-pipe utilisation of the step's MFMA kernels from SQ_VALU_MFMA_BUSY_CYCLES (own PMC pass, kernel trace only).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_mfma_util
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export CORRIF_SERIAL=1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES -d $O/pmc -o m --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > $O/run.log 2>&1 || echo "pmc pass failed"
ls $O/pmc | head
