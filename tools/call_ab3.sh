#!/bin/bash
# schedule knobs re-checked with the split-bf16 kernels
set -o pipefail
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
run() { local tag=$1; shift; env "$@" timeout -k 10 300 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-kernel-timing > gpurun_out/ab3_$tag.log 2>&1; local rc=$?; echo "$tag rc=$rc $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/ab3_$tag.log)"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi; }
run base A=1
run lanes3 CORRIF_DECODER_SPLIT=3
run lanes1 CORRIF_DECODER_SPLIT=1
run sidewgrad CORRIF_SIDE_WGRAD=1
run nointerleave CORRIF_INTERLEAVE=0
run nobwdstats CORRIF_BWD_STATS=0
run base2 A=1
