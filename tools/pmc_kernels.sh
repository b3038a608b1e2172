#!/bin/bash
# Dev probe (GPU box): average PMC counter values per kernel for a python probe.
#   tools/pmc_kernels.sh "SQ_INSTS_VALU SQ_INSTS_SALU" tools/resolve_blocks.py child
export TMPDIR=/tmp
REPO=$PWD
D=/tmp/pmc_$$
cd /tmp
rocprofv3 --pmc $1 --output-format csv -d $D -- python3 "$REPO/$2" "${@:3}" > /dev/null 2> $D.err || tail -5 $D.err
python3 - "$D" <<'PY'
import csv, glob, sys, collections
fs = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")
if not fs:
    print("no counter output"); sys.exit(0)
by = collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    by[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(by.items()):
    if len(v) > 20 and not k.startswith("__amd"):
        v = v[-40:]
        print("%-26s %-22s avg %14.1f" % (k[:26], c, sum(v) / len(v)))
PY
