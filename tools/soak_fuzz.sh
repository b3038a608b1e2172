#!/bin/bash
# SPDX-License-Identifier: GPL-3.0-or-later
# Fuzz soak that keeps what a failure says (round 3 kept a summary line and lost a wholesale failure's cause).
#
#   tools/soak_fuzz.sh TAG [PROCESSES] [SEEDS] [MEDIUM] [LONG]
#
# Runs PROCESSES fresh python processes of tests/test_gpu_fuzz.py (and the grouped-candidates cases of test_gpu_tail_groups.py), one after the other, each with -x --tb=short, the
# library's self-test trace on and tests/_diag.py's route differential on a mismatch (same ROM + plan scanned again in
# the same process through every route, header words, health counters, ROM dumped under gpurun_out/fuzz_failures/).
# Everything a process prints is kept in gpurun_out/soak/TAG_pN.log; one line per process goes to
# gpurun_out/soak/summary.tsv (copied into profiles/rNN_fuzz_soak.tsv by hand).
# Run on the GPU box through gpurun, program first -- no env/bash -c wrappers around a GPU process are needed here
# (no profiler): gpurun -- 'bash tools/soak_fuzz.sh lease7 3 1000 100 100'
set -u
TAG=${1:?tag}
PROCS=${2:-1}
SEEDS=${3:-3000}
MEDIUM=${4:-300}
LONG=${5:-300}
OUT=gpurun_out/soak
mkdir -p "$OUT"
BOX="$(hostname 2>/dev/null)-$(cat /proc/sys/kernel/random/boot_id 2>/dev/null | cut -c1-8)"
for p in $(seq 1 "$PROCS"); do
   LOG="$OUT/${TAG}_p${p}.log"
   T0=$(date +%s)
   MM_FUZZ_SEEDS=$SEEDS MM_FUZZ_MEDIUM=$MEDIUM MM_FUZZ_LONG=$LONG MMOORE_TRACE=selftest \
      timeout 3000 python3 -m pytest tests/test_gpu_tail_groups.py tests/test_gpu_fuzz.py -m gpu -x --tb=short -q -p no:cacheprovider -s \
      -k "grouped_candidates or against_oracle or medium_roms or long_keywords or left_the_routes" >"$LOG" 2>&1
   RC=$?
   T1=$(date +%s)
   LAST=$(grep -E "passed|failed|error" "$LOG" | tail -1 | tr '\t' ' ')
   HEALTH=$(grep -E "route health after the fuzz" "$LOG" | tail -1 | tr '\t' ' ')
   SELF=$(grep -E "self-test" "$LOG" | tail -1 | tr '\t' ' ')
   printf "%s\t%s\tp%s\tseeds=%s/%s/%s\trc=%s\t%ss\t%s\t%s\t%s\n" "$TAG" "$BOX" "$p" "$SEEDS" "$MEDIUM" "$LONG" "$RC" "$((T1 - T0))" "$LAST" "$SELF" "$HEALTH" >>"$OUT/summary.tsv"
   # a passing process's log is dots: keep its head and tail only
   if [ "$RC" = "0" ]; then
      { head -c 2000 "$LOG"; echo; echo "[...]"; tail -c 3000 "$LOG"; } >"$LOG.short" && mv "$LOG.short" "$LOG"
   fi
done
tail -n "$PROCS" "$OUT/summary.tsv"
