#!/bin/bash
# Dev probe (GPU box): SQ counters of the forward engine per launch and case, in separate rocprofv3 --pmc passes
# (tools/forward_one.py: 30 forced scans of 1 GiB per case).   tools/forward_counters.sh [CASE ...] > profiles/r06_forward_sq_counters.txt
cd "$(dirname "$0")/.."
for c in "${@:-flood4096 qz alpha16 plain8}"; do
   for case in $c; do
      echo "== python3 tools/forward_one.py $case"
      python3 tools/forward_one.py "$case" | tail -1
      tools/pmc_kernels.sh "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT" tools/forward_one.py "$case" | grep "mm_forward"
      tools/pmc_kernels.sh "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" tools/forward_one.py "$case" | grep "mm_forward"
   done
done
