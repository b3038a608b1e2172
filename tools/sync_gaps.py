# SPDX-License-Identifier: GPL-3.0-or-later
"""Dev probe: from a rocprofv3 kernel trace of tools/scan_probe.bin (one scan at a time) -- the gaps between the kernels of
consecutive scans: streaming kernel end -> tail start, tail end -> next streaming kernel start."""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith(("void mm_filter", "void mm_scan_tail"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
f2t, t2f, fd, td = [], [], [], []
for a, b in zip(rows, rows[1:]):
    gap = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
    if "filter" in a["Kernel_Name"] and "tail" in b["Kernel_Name"]:
        f2t.append(gap)
    if "tail" in a["Kernel_Name"] and "filter" in b["Kernel_Name"]:
        t2f.append(gap)
for r in rows:
    (fd if "filter" in r["Kernel_Name"] else td).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
h = len(f2t) // 2
avg = lambda v: sum(v[h:]) / max(1, len(v[h:]))
print("%s: streaming %.1f us, gap %.1f us, tail %.1f us, gap to the next scan's streaming kernel %.1f us (second half of %d scans)" % (
    sys.argv[2] if len(sys.argv) > 2 else "", avg(fd), avg(f2t), avg(td), avg(t2f), len(fd)))
