#!/bin/bash
# PMC counters of the split-loop lab kernels (own pass, kernel trace only)
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_lab
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $O/p -o l --output-format csv -- $R/tools/probe/split_lab.bin > $O/run.log 2>&1 || echo "pmc failed"
python3 - <<'PY'
import csv, collections, os
p=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/pmc_lab/p/l_counter_collection.csv'
agg=collections.defaultdict(lambda: collections.Counter())
dur=collections.Counter(); n=collections.Counter()
for r in csv.DictReader(open(p)):
    k=r['Kernel_Name'].split('(')[0].replace('void ','')
    agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
for k,c in agg.items():
    if 'gemm' not in k: continue
    w=c.get('SQ_WAVE_CYCLES',1)
    print('%-46s mfma_busy/busy %.3f | per wave-cycle: wait_lds %.3f wait_any %.3f active_lds %.3f active_valu %.3f | bank_conflict/active_lds %.3f' % (k[:46], c['SQ_VALU_MFMA_BUSY_CYCLES']/max(c['SQ_BUSY_CYCLES'],1), c['SQ_WAIT_INST_LDS']/w, c['SQ_WAIT_INST_ANY']/w, c['SQ_ACTIVE_INST_LDS']/w, c['SQ_ACTIVE_INST_VALU']/w, c['SQ_LDS_BANK_CONFLICT']/max(c['SQ_ACTIVE_INST_LDS'],1)))
PY
