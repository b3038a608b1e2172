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


def counter_avgs(dirname, counter):
    per = {}
    for f in glob.glob(os.path.join(out, dirname, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            name = row["Kernel_Name"].split("(")[0]
            per.setdefault(name, []).append(float(row["Counter_Value"]))
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
lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "monkey-moore_amd", "lib", "libmmoore_hip.so")
summary = {
    # bench.py reports `roofline.traffic` from this file only while the library it runs is THIS build
    "library_sha16": hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16],
    "command_fetch": "rocprofv3 --pmc FETCH_SIZE --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-depth --prewarm-s 0.05",
    "command_write": "rocprofv3 --pmc WRITE_SIZE --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-depth --prewarm-s 0.05",
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
