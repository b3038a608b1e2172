#!/bin/bash
# Dev probe (GPU box): SQ counters of the streaming kernels, per launch, in separate rocprofv3 --pmc passes (eight SQ
# counters at most per pass) -- one launch over the whole 4 GiB ROM per scan (tools/keyword_kernel_ms.py).
#   tools/filter_counters.sh [keyword[:elem] ...] > profiles/r06_filter_sq_counters.txt
cd "$(dirname "$0")/.."
for kw in "${@:-relativesrch qz**mb textsrch:2}"; do
   for k in $kw; do
      echo "== python3 tools/keyword_kernel_ms.py $k (one streaming launch per scan; averages over the launches of each kernel)"
      tools/pmc_kernels.sh "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_FLAT" tools/keyword_kernel_ms.py $k | grep "mm_filter\|mm_scan_tail"
      tools/pmc_kernels.sh "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" tools/keyword_kernel_ms.py $k | grep "mm_filter\|mm_scan_tail"
   done
done
