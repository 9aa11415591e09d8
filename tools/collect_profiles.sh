#!/bin/bash
# Round profiles (GPU box): kernel stats of the default (concurrent) and single-stream step, and the HBM traffic counters
# (FETCH_SIZE / WRITE_SIZE in separate passes, no other trace domains).  Output under gpurun_out/prof_<tag>/.
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing"
rocprofv3 --kernel-trace --stats -d $O/concurrent -o c --output-format csv -- python $R/bench.py $ARGS > $O/concurrent.log 2>&1 || echo "concurrent failed"
export CORRIF_SERIAL=1
rocprofv3 --kernel-trace --stats -d $O/serial -o s --output-format csv -- python $R/bench.py $ARGS > $O/serial.log 2>&1 || echo "serial failed"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o f --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > $O/fetch.log 2>&1 || echo "fetch failed"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o w --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > $O/write.log 2>&1 || echo "write failed"
rm -f $O/*/*_kernel_trace.csv.bak
ls -la $O/*
