#!/bin/bash
# PMC passes over tools/conv_microbench.py (GPU box).  Output: gpurun_out/pmc_conv/pass*/
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export REPS=2 NSHAPES=3
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY" \
         "TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" \
         "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/pmc_conv/pass$i -o p --output-format csv -- python $R/tools/conv_microbench.py > $R/gpurun_out/pmc_conv_pass$i.log 2>&1 || echo "pass $i failed"
done
ls $R/gpurun_out/pmc_conv/*
