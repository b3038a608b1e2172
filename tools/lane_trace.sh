#!/bin/bash
# SPDX-License-Identifier: GPL-3.0-or-later
# Dev probe: where the kernels of two scans in flight lie in time, by the library's own HIP events
# (MMOORE_LANE_TRACE) and by rocprofv3's kernel trace of the same kind of run.
export TMPDIR=/tmp
REPO=$PWD
STEPS=${1:-200}
MMOORE_LANE_TRACE=1 python3 bench.py --steps $STEPS --warmup 10 --no-cpu-baseline --no-other-depth 2>&1 | grep "^lane\|^{" | tail -13 | cut -c1-400
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/lt -- python3 $REPO/bench.py --steps $STEPS --warmup 10 --no-cpu-baseline --no-other-depth > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob("/tmp/lt/**/*kernel_trace.csv", recursive=True):
    rows += [r for r in csv.DictReader(open(f)) if "mm_filter" in r["Kernel_Name"] or "mm_scan_tail" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[-24]["Start_Timestamp"])
for r in rows[-24:]:
    print("%-14s stream %s queue %s  %.1f .. %.1f us (%.1f)" % (r["Kernel_Name"][5:18], r["Stream_Id"], r["Queue_Id"], (int(r["Start_Timestamp"]) - t0) / 1e3,
          (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
