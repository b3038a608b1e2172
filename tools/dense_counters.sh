#!/bin/bash
# Dev probe (GPU box, round 5): SQ counters of the streaming kernel per launch on the text-like ROM, a keyword without and
# keywords with candidates, in separate rocprofv3 --pmc passes.   tools/dense_counters.sh > profiles/r05_dense_sq_counters.txt
cd "$(dirname "$0")/.."
for kw in relativesrch water 'th*s'; do
   echo "== python3 tools/dense_one.py '$kw'"
   python3 tools/dense_one.py "$kw" | tail -1
   tools/pmc_kernels.sh "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM" tools/dense_one.py "$kw" | grep "mm_filter"
   tools/pmc_kernels.sh "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS" tools/dense_one.py "$kw" | grep "mm_filter"
   tools/pmc_kernels.sh "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT" tools/dense_one.py "$kw" | grep "mm_filter"
done
