# SPDX-License-Identifier: GPL-3.0-or-later
"""Dev probe: from `rocprofv3 --kernel-trace --hip-runtime-trace` of tools/scan_probe.bin -- per scan, when the streaming
kernel's launch call began / returned and when the kernel started, when the tail kernel ended and when the NEXT scan's launch began."""
import csv, glob, sys
k, api = [], []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    k += [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith(("void mm_filter", "void mm_scan_tail"))]
for f in glob.glob(sys.argv[1] + "/**/*hip_api_trace.csv", recursive=True):
    api += [r for r in csv.DictReader(open(f)) if "LaunchKernel" in r["Function"]]
k.sort(key=lambda r: int(r["Start_Timestamp"]))
api.sort(key=lambda r: int(r["Start_Timestamp"]))
filt = [r for r in k if "filter" in r["Kernel_Name"]]
tail = [r for r in k if "tail" in r["Kernel_Name"]]
# launches come in pairs (filter, tail) per scan, preceded by set-up kernels: align from the end
n = min(len(filt), len(tail), len(api) // 2)
filt, tail, api = filt[-n:], tail[-n:], api[-2 * n:]
d1, d2, d3 = [], [], []
for i in range(n // 2, n):
    call0, call1 = int(api[2 * i]["Start_Timestamp"]), int(api[2 * i]["End_Timestamp"])
    d1.append((int(filt[i]["Start_Timestamp"]) - call0) / 1e3)
    d2.append((int(filt[i]["Start_Timestamp"]) - call1) / 1e3)
    if i + 1 < n:
        d3.append((int(api[2 * i + 2]["Start_Timestamp"]) - int(tail[i]["End_Timestamp"])) / 1e3)
avg = lambda v: sum(v) / max(1, len(v))
print("%s: streaming kernel starts %.1f us after its launch call began (%.1f after it returned); the next scan's launch call begins %.1f us after the tail kernel ended" % (
    sys.argv[2] if len(sys.argv) > 2 else "", avg(d1), avg(d2), avg(d3)))
