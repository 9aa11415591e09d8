#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python tools/mem_probe.py 16 8 256 > gpurun_out/r3o_mem_b16.log 2>&1; echo "rc=$?" >> gpurun_out/r3o_mem_b16.log
grep "GB per step\|Error\|rc=" gpurun_out/r3o_mem_b16.log | cut -c1-300
