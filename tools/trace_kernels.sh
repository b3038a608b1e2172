#!/bin/bash
# Dev probe (GPU box): per-kernel durations of a python probe under rocprofv3 --kernel-trace.
#   tools/trace_kernels.sh tools/resolve_blocks.py child
export TMPDIR=/tmp
REPO=$PWD
D=/tmp/prof_$$
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $D -- python3 "$REPO/$1" "${@:2}" > /dev/null 2>&1
python3 - "$D" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
by = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    by[r["Kernel_Name"].split("(")[0]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in by.items():
    if len(v) > 20 and not k.startswith("__amd"):
        h = len(v) // 2
        a, b = v[:h][-30:], v[h:][-30:]
        print("%-26s n=%-4d first half avg %7.1f us (min %.1f) | second half avg %7.1f us (min %.1f)" % (
            k[:26], len(v), sum(a) / len(a) / 1e3, min(a) / 1e3, sum(b) / len(b) / 1e3, min(b) / 1e3))
PY
