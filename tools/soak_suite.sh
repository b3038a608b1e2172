#!/bin/bash
# SPDX-License-Identifier: GPL-3.0-or-later
# The driver's own GPU command on N fresh leases (one gpurun call = one fresh MI355X box), one ledger line each:
#   tools/soak_suite.sh TAG N            -> profiles/r06_suite_soak.tsv
# (the suite must be green by construction -- no wall-clock gate -- on every box, the slow ones included.)
TAG=${1:?tag}; N=${2:-6}
LEDGER=${LEDGER:-profiles/r06_suite_soak.tsv}
for i in $(seq 1 "$N"); do
   for attempt in 1 2 3 4 5 6; do
      /usr/local/graft/bin/gpurun --timeout 2400 -- 'S=$(date +%s); python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > gpurun_out/soak_suite.log; E=$(date +%s); echo "LEDGER $(hostname)-$(cat /proc/sys/kernel/random/boot_id | cut -c1-8) $((E-S))s $(tail -1 gpurun_out/soak_suite.log)"' > gpurun_out/soak_suite_$i.out 2>&1
      rc=$?
      if [ "$rc" != "2" ] && [ "$rc" != "3" ]; then break; fi
      sleep 45
   done
   line=$(grep -h "^LEDGER" gpurun_out/soak_suite_$i.out | sed 's/^LEDGER //')
   printf '%s\t%s\tpython -m pytest tests -x -q -m gpu\tgpurun rc=%s\t%s\n' "$TAG" "$i" "$rc" "$line" >> "$LEDGER"
   tail -1 "$LEDGER"
done
