#!/bin/bash
# Dev probe (GPU box): SQ counters of the streaming kernels, per launch, in separate rocprofv3 --pmc
# passes (eight SQ counters at most per pass).   tools/filter_counters.sh > profiles/r05_filter_sq_counters.txt
cd "$(dirname "$0")/.."
for cfg in C2 C4; do
   echo "== $cfg: python3 tools/sync_scans.py $cfg 30 (one streaming launch per scan; averages over the last launches of each kernel)"
   tools/pmc_kernels.sh "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_FLAT" tools/sync_scans.py $cfg 30 | grep "mm_filter\|mm_scan_tail"
   tools/pmc_kernels.sh "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" tools/sync_scans.py $cfg 30 | grep "mm_filter\|mm_scan_tail"
   tools/pmc_kernels.sh "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" tools/sync_scans.py $cfg 30 | grep "mm_filter\|mm_scan_tail"
done
