#!/bin/bash
# SPDX-License-Identifier: GPL-3.0-or-later
# The driver's own GPU command on N fresh leases (one gpurun call = one fresh MI355X box), one ledger line each:
#   tools/soak_suite.sh TAG N            -> profiles/r06_suite_soak.tsv
# (the suite must be green by construction -- no wall-clock gate -- on every box, the slow ones included.)
TAG=${1:?tag}; N=${2:-6}
LEDGER=${LEDGER:-profiles/r06_suite_soak.tsv}
for i in $(seq 1 "$N"); do
   for attempt in 1 2 3 4 5 6; do
      # (pytest's summary line is not the last one: RCCL prints its version banner when the process ends)
      /usr/local/graft/bin/gpurun --timeout 2400 -- 'S=$(date +%s); python -m pytest tests -x -q -m gpu > gpurun_out/soak_suite.log 2>&1; RC=$?; E=$(date +%s); echo "LEDGER $(hostname)-$(cat /proc/sys/kernel/random/boot_id | cut -c1-8) pytest rc=$RC $((E-S))s $(grep -E " passed| failed| error" gpurun_out/soak_suite.log | tail -1)"; grep -E "^FAILED|^ERROR" gpurun_out/soak_suite.log | head -5' > gpurun_out/soak_suite_$i.out 2>&1
      rc=$?
      if [ "$rc" != "2" ] && [ "$rc" != "3" ]; then break; fi
      sleep 45
   done
   line=$(grep -h "^LEDGER" gpurun_out/soak_suite_$i.out | sed 's/^LEDGER //')
   printf '%s\t%s\tpython -m pytest tests -x -q -m gpu\tgpurun rc=%s\t%s\n' "$TAG" "$i" "$rc" "$line" >> "$LEDGER"
   tail -1 "$LEDGER"
done
