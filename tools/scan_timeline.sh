#!/bin/bash
# Dev probe (GPU box): the kernels of the LAST synchronous scan of a keyword on the 4 GiB ROM, in time order -- start (us
# after the first of them), duration, grid, name -- under rocprofv3 --kernel-trace.   tools/scan_timeline.sh abcd [ROWS]
export TMPDIR=/tmp
REPO=$PWD
D=/tmp/prof_$$
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $D -- python3 "$REPO/tools/keyword_sweep.py" "$1" > /dev/null 2>&1
python3 - "$D" "${2:-40}" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))[-int(sys.argv[2]):]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    print("%9.1f us  +%8.1f us  grid %8s  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                               r.get("Grid_Size_X", r.get("Grid_Size", "?")), r["Kernel_Name"].split("(")[0][:70]))
PY
