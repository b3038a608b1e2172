#!/bin/bash
# SPDX-License-Identifier: GPL-3.0-or-later
# Fresh-lease repeats of the fuzz soak, from the build container: every repeat is one gpurun call = one fresh MI355X box
# and one fresh process (the round-3 failure was a property of a process or a box, not of a seed).
#   tools/soak_leases.sh FIRST LAST [SEEDS MEDIUM LONG]
# Lines land in gpurun_out/soak/summary.tsv (merged back by gpurun); copy them to profiles/rNN_fuzz_soak.tsv.
FIRST=${1:?first}; LAST=${2:?last}; SEEDS=${3:-600}; MEDIUM=${4:-60}; LONG=${5:-60}
for i in $(seq "$FIRST" "$LAST"); do
   for attempt in 1 2 3 4 5 6; do
      /usr/local/graft/bin/gpurun --timeout 1500 -- "bash tools/soak_fuzz.sh r04_lease$i 1 $SEEDS $MEDIUM $LONG" >gpurun_out/soak_lease_$i.out 2>&1
      rc=$?
      # 2 = refused (another call is running), 3 = no box free: wait and try again
      if [ "$rc" != "2" ] && [ "$rc" != "3" ]; then break; fi
      sleep 45
   done
   tail -2 gpurun_out/soak_lease_$i.out
   # (gpurun_out/soak/summary.tsv is rewritten by every call: the ledger lives in profiles/)
   grep -h "^r04_lease$i\b" gpurun_out/soak_lease_$i.out | python3 -c "
import re, sys
for ln in sys.stdin:
    c = ln.rstrip('\n').split('\t')
    m = re.search(r'route health after the fuzz: (\{.*\})', c[8] if len(c) > 8 else '')
    print('\t'.join(c[:8] + [m.group(1) if m else '']))" >> "${LEDGER:-profiles/r04_fuzz_soak.tsv}"
done
