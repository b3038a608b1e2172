#!/bin/bash
# Dev probe (GPU box): SQ counters of the forward engine per launch and case, in separate
# rocprofv3 --pmc passes (tools/forward_one.py: 30 forced scans of 1 GiB per case).
#   tools/forward_counters.sh > profiles/r05_forward_sq_counters.txt
cd "$(dirname "$0")/.."
for c in "MMOORE_FORWARD_SWEEP=0 flood4096" "MMOORE_FORWARD_SWEEP=0 alpha16" "X=1 flood4096" "X=1 plain8" "X=1 plain16" "X=1 long41"; do
   set -- $c
   echo "== $1 python3 tools/forward_one.py $2"
   env $1 python3 tools/forward_one.py $2 | tail -1
   env $1 tools/pmc_kernels.sh "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT" tools/forward_one.py $2 | grep "mm_forward"
   env $1 tools/pmc_kernels.sh "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" tools/forward_one.py $2 | grep "mm_forward"
done
