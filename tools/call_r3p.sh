#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
step() { local log=$1 secs=$2; shift 2; timeout -k 10 $secs "$@" > gpurun_out/$log 2>&1; local rc=$?; echo "rc=$rc" >> gpurun_out/$log; echo "== $log rc=$rc: $(tail -2 gpurun_out/$log | cut -c1-900 | tr '\n' ' ')"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi; }
B="python bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-kernel-timing"
step r3p_cfg_b32_d8_256.log 300 $B --batch 32 --bands 8 --size 256
step r3p_cfg2_b64_d8_256.log 400 $B --batch 64 --bands 8 --size 256
CORRIF_GROUPED=0 step r3p_cfg2_twins.log 400 $B --batch 64 --bands 8 --size 256
