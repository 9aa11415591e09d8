#!/bin/bash
# Round profiles (GPU box): kernel stats of the default (concurrent) and single-stream step, the HBM traffic counters
# (FETCH_SIZE / WRITE_SIZE in separate passes, no other trace domains) and the matrix-pipe busy counter.  Output under gpurun_out/prof_<tag>/;
# the reduced summaries are written next to it, ready to be copied into profiles/.
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/concurrent -o c --output-format csv -- python $R/bench.py $ARGS > $O/concurrent.log 2>&1 || echo "concurrent failed"
python $R/tools/timeline.py $O/concurrent/c_kernel_trace.csv 244 > $O/timeline.txt 2>&1
export CORRIF_SERIAL=1
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/serial -o s --output-format csv -- python $R/bench.py $ARGS > $O/serial.log 2>&1 || echo "serial failed"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o f --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > $O/fetch.log 2>&1 || echo "fetch failed"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o w --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > $O/write.log 2>&1 || echo "write failed"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES -d $O/pmc -o m --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > $O/mfma.log 2>&1 || echo "mfma pmc failed"
python $R/tools/reduce_traffic.py $O 2 $O/serial/s_kernel_stats.csv 3 > $O/${TAG}_mfma_traffic.json 2> $O/reduce_traffic.err
python $R/tools/reduce_mfma_util.py $O/pmc/m_counter_collection.csv > $O/${TAG}_mfma_util.json 2> $O/reduce_util.err
rm -f $O/*/*_kernel_trace.csv $O/*/*_counter_collection.csv
ls -la $O $O/serial | head -30
