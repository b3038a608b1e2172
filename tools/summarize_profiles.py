# SPDX-License-Identifier: GPL-3.0-or-later
"""Turns the raw rocprofv3 output of tools/collect_profiles.sh into the small files kept under
profiles/: <tag>_kernel_stats.csv, <tag>_pmc_summary.json, <tag>_bench_n1.json."""
import csv
import glob
import json
import os
import shutil
import sys

out, tag = sys.argv[1], sys.argv[2]
dst = os.path.join(os.path.dirname(out), "profiles_" + tag)
os.makedirs(dst, exist_ok=True)

stats = glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(dst, tag + "_kernel_stats.csv"))
shutil.copy(os.path.join(out, "bench_n1.json"), os.path.join(dst, tag + "_bench_n1.json"))
stats1 = glob.glob(os.path.join(out, "stats_depth1", "**", "*kernel_stats.csv"), recursive=True)
if stats1:
    shutil.copy(stats1[0], os.path.join(dst, tag + "_kernel_stats_depth1.csv"))
for cfg in ("C3", "C4", "C4BE"):
    sc = glob.glob(os.path.join(out, "stats_depth1_" + cfg, "**", "*kernel_stats.csv"), recursive=True)
    if sc:
        shutil.copy(sc[0], os.path.join(dst, "%s_kernel_stats_depth1_%s.csv" % (tag, cfg)))
if os.path.exists(os.path.join(out, "bench_n1_driver_shape.json")):
    shutil.copy(os.path.join(out, "bench_n1_driver_shape.json"), os.path.join(dst, tag + "_bench_n1_driver_shape.json"))


def by_phase():
    """The default run's kernel trace cut into bench.py's phases (launch order): the pre-warm scans and the
    synchronous leg run one scan at a time, the warm-up + timed steps two in flight.  Kept next to the trace
    itself so that the figures can be recomputed."""
    import gzip
    traces = glob.glob(os.path.join(out, "stats", "**", "*kernel_trace.csv"), recursive=True)
    line = [l for l in open(os.path.join(out, "bench_under_rocprof.json")).read().splitlines() if l.startswith("{")]
    if not traces or not line:
        return
    b = json.loads(line[-1])
    rows = []
    for f in traces:
        rows += [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith(("void mm_filter", "void mm_scan_tail"))]
    if not rows:
        return
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    with gzip.open(os.path.join(dst, tag + "_kernel_trace_default_run.csv.gz"), "wt") as g:
        w = csv.writer(g)
        w.writerow(["Kernel_Name", "Stream_Id", "Start_Timestamp", "End_Timestamp"])
        for r in rows:
            w.writerow([r["Kernel_Name"].split("(")[0], r["Stream_Id"], r["Start_Timestamp"], r["End_Timestamp"]])
    filt = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "mm_filter" in r["Kernel_Name"]]
    tail = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "mm_scan_tail" in r["Kernel_Name"]]
    K, W, pre = b["steps"], b["warmup"], b["config"]["prewarm_scans"]
    depth = b["config"]["scans_in_flight"]
    phases = [("prewarm, one scan at a time", pre), ("warm-up, %d ticket(s) outstanding" % depth, W),
              ("TIMED steps, %d ticket(s) outstanding" % depth, K),
              ("warm-up of the other leg", max(W, 4)), ("other leg, %d ticket(s) outstanding" % (1 if depth > 1 else 3), K)]
    if b["config"].get("launches_by_phase"):
        # round 5: bench.py counts the streaming launches of every phase itself (a synchronous scan of a big ROM is a
        # pipeline of parts: more than one launch per scan)
        phases = [(name, int(n)) for name, n in b["config"]["launches_by_phase"]]
    res, at = {"launches_of_the_streaming_kernel": len(filt), "phases": []}, 0
    for name, n in phases:
        f, t = filt[at:at + n], tail[at:at + n]
        at += n
        if len(f) < 2:
            continue
        res["phases"].append({
            "phase": name, "launches": len(f),
            "streaming_kernel_avg_us": sum(e - s for s, e in f) / len(f) / 1e3,
            "streaming_kernel_start_to_start_avg_us": (f[-1][0] - f[0][0]) / (len(f) - 1) / 1e3,
            "tail_kernel_avg_us (dispatch to end: with two in flight it waits for registers beside the other lane's streaming kernel)":
                sum(e - s for s, e in t) / max(len(t), 1) / 1e3,
        })
    res["bench_line_of_this_run"] = {"ms_per_step": b["ms_per_step"], "roofline_kernel_ms": b["roofline"]["kernel_ms"],
                                     "timed_region": b["roofline"].get("timed_region")}
    json.dump(res, open(os.path.join(dst, tag + "_kernel_trace_by_phase.json"), "w"), indent=1)
    print(json.dumps(res, indent=1))


by_phase()


def counter_avgs(dirname, counter):
    per = {}
    for f in glob.glob(os.path.join(out, dirname, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            name = row["Kernel_Name"].split("(")[0]
            per.setdefault(name, []).append(float(row["Counter_Value"]))
    # (round 6: the library's device warm-up at mmh_create launches the streaming kernels on an 8 MiB ROM -- only the
    # launches over the bench ROM count: everything within half of the kernel's largest value)
    per = {k: [x for x in v if x >= 0.5 * max(v)] for k, v in per.items()}
    return {k: (sum(v) / len(v), len(v)) for k, v in per.items()}


stats_cfg = glob.glob(os.path.join(out, "stats_configs", "**", "*kernel_stats.csv"), recursive=True)
if stats_cfg:
    shutil.copy(stats_cfg[0], os.path.join(dst, tag + "_config_kernel_stats.csv"))
if os.path.exists(os.path.join(out, "config_times.log")):
    shutil.copy(os.path.join(out, "config_times.log"), os.path.join(dst, tag + "_config_times.log"))

fetch, write = counter_avgs("pmc_fetch", "FETCH_SIZE"), counter_avgs("pmc_write", "WRITE_SIZE")
bench = json.loads(open(os.path.join(out, "bench_n1.json")).read().strip().splitlines()[-1])
kernel = "void " + bench["roofline"]["kernel"]
alg = bench["roofline"]["algorithmic_bytes"]
import hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
device_source_sha16 = load_package().build.device_source_sha16
lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "monkey-moore_amd", "lib", "libmmoore_hip.so")
summary = {
    # bench.py reports `roofline.traffic` from this file only while the library's sources are THESE
    "device_source_sha16": device_source_sha16(),
    "library_sha16": hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16],
    "command_fetch": "rocprofv3 --pmc FETCH_SIZE --output-format csv -- python3 bench.py --steps 3 --warmup 1 --depth 1 --no-split --no-cpu-baseline --no-other-depth --no-other-configs --no-strong --no-read-probe --prewarm-s 0.05",
    "command_write": "rocprofv3 --pmc WRITE_SIZE --output-format csv -- python3 bench.py --steps 3 --warmup 1 --depth 1 --no-split --no-cpu-baseline --no-other-depth --no-other-configs --no-strong --no-read-probe --prewarm-s 0.05",
    "note": "separate --pmc passes; counters are in KiB; per MI355X_MICROARCH.md gfx950 FETCH_SIZE reports half of the bytes of a "
            "wide coalesced (16 B/lane) streaming read, so the read side is doubled; WRITE_SIZE is exact "
            "(calibration: mm_synth_fill writes the whole ROM and reports exactly its size)",
    "kernel": kernel,
    "per_kernel": {},
}
for k in sorted(set(fetch) | set(write)):
    summary["per_kernel"][k] = {
        "FETCH_SIZE_KiB_avg_per_launch": fetch.get(k, (None, 0))[0], "FETCH_SIZE_launches": fetch.get(k, (None, 0))[1],
        "WRITE_SIZE_KiB_avg_per_launch": write.get(k, (None, 0))[0], "WRITE_SIZE_launches": write.get(k, (None, 0))[1],
    }
if kernel in fetch and kernel in write:
    raw = fetch[kernel][0] * 1024.0
    wr = write[kernel][0] * 1024.0
    summary.update({
        "fetch_size_bytes_raw": raw, "fetch_bytes_corrected_x2": 2 * raw, "write_bytes": wr,
        "hbm_traffic_bytes_per_launch": 2 * raw + wr, "algorithmic_bytes_per_launch": alg,
        "traffic_over_algorithmic": (2 * raw + wr) / alg,
    })
json.dump(summary, open(os.path.join(dst, tag + "_pmc_summary.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != "per_kernel"}, indent=1))
if stats:
    print(open(stats[0]).read()[:1500])
