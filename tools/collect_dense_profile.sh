#!/bin/bash
# Dev probe (GPU box): kernel stats + SQ counters of the forward (dense) engine on 1 GiB.
#   tools/collect_dense_profile.sh r02a
set -u
TAG=${1:-r02}
REPO=$PWD
OUT=$REPO/gpurun_out/dense_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for W in plain wild; do
  python3 "$REPO/tools/dense_profile.py" $W 6 > "$OUT/times_$W.log" 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$W" -- python3 "$REPO/tools/dense_profile.py" $W 24 > /dev/null 2> "$OUT/stats_$W.err"
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$OUT/pmc1_$W" -- python3 "$REPO/tools/dense_profile.py" $W 24 > /dev/null 2> "$OUT/pmc1_$W.err"
  rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$OUT/pmc2_$W" -- python3 "$REPO/tools/dense_profile.py" $W 24 > /dev/null 2> "$OUT/pmc2_$W.err"
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM --output-format csv -d "$OUT/pmc3_$W" -- python3 "$REPO/tools/dense_profile.py" $W 24 > /dev/null 2> "$OUT/pmc3_$W.err"
done
cd "$REPO"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, os
out = sys.argv[1]
for W in ("plain", "wild"):
    print("=====", W, open(os.path.join(out, "times_%s.log" % W)).read().strip())
    for f in glob.glob(os.path.join(out, "stats_" + W, "**", "*kernel_stats.csv"), recursive=True):
        print(open(f).read()[:2500])
    by = collections.defaultdict(list)
    for d in ("pmc1_", "pmc2_", "pmc3_"):
        for f in glob.glob(os.path.join(out, d + W, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                by[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(by.items()):
        if "mm_dense" in k or "mm_forward" in k:
            v = v[-16:]
            print("%-34s %-22s n=%-3d avg %16.1f" % (k[:34], c, len(v), sum(v) / len(v)))
PY
