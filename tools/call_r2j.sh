mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r2j
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/conc -o c --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $O/conc.log 2>&1 || echo "conc prof failed"
grep "^{" $O/conc.log | cut -c60-150
python $R/tools/timeline.py $O/conc/c_kernel_trace.csv 300 > $O/timeline.txt 2>&1
rm -f $O/conc/c_kernel_trace.csv
cat $O/timeline.txt | cut -c1-220
