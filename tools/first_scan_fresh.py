# SPDX-License-Identifier: GPL-3.0-or-later
"""Dev probe: the FIRST mmh_scan of a process against its steady state, per keyword class -- a child
process per keyword: context, 4 GiB ROM (C2's recipe) allocated and filled, then scans 1..10 timed on the host clock.
What the first scan still pays that a later one does not is what the library set up lazily.
    python tools/first_scan_fresh.py [keyword[:elem] ...]      -> profiles/r06_first_scan.log"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT = ["relativesrch", "re*ative*ear*hxy", "qzvk", "qzv", "qz*k", "qz**mb", "q**k**x", "ab*de", "a*cd*f", "aaaa", "abcd", "qz", "q*v",
           "q" * 2 + "zvkmbxw" * 9, "textsrch:2", "q*v*m:2", "qz:2"]


def child(item):
    import numpy as np
    sys.path.insert(0, ROOT)
    from __graft_entry__ import load_package
    mm = load_package()
    kw, _, elem = item.partition(":")
    elem = int(elem or 1)
    N, BLOCK = 4 << 30, 524288
    eng = mm.Engine(0)
    eng.alloc(N)
    mm.synth.RomSpec(42, N, "relativesrch", 1, None, False, BLOCK).apply_device(eng)
    eng.download(0, 16)                                       # (the ROM's set-up is over: nothing of it is in the first scan's time)
    plan = mm.plan_relative(elem, kw, ord("*") if "*" in kw else 0)
    wall, n = [], 0
    for i in range(10):
        t0 = time.perf_counter()
        r = eng.scan(plan, block_bytes=BLOCK, cap=1 << 25)
        wall.append((time.perf_counter() - t0) * 1e3)
        n = len(r)
    steady = float(np.median(wall[3:]))
    ctr, tm = eng.counters(), eng.timings()
    # the same scan after the device has sat idle for half a second (clocks down): what part of "first" is just "after a pause"
    idle = []
    for _ in range(3):
        time.sleep(0.5)
        t0 = time.perf_counter()
        eng.scan(plan, block_bytes=BLOCK, cap=1 << 25)
        idle.append((time.perf_counter() - t0) * 1e3)
    print("%2d-bit %-18s %9d matches path %d parts %d | first %8.3f ms, second %8.3f, steady (median of 4..10) %8.3f | first / steady %.2f | after 0.5 s idle %8.3f ms (%.2f x steady)" % (
        8 * elem, kw[:18], n, ctr["path"], tm.get("parts", 0), wall[0], wall[1], steady, wall[0] / steady, float(np.median(idle)), float(np.median(idle)) / steady), flush=True)


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        print("# first scan of a fresh process (context + 4 GiB ROM set up before the clock starts) against its steady state, host clock around mmh_scan")
        for item in (sys.argv[1:] or DEFAULT):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", item], capture_output=True, text=True, timeout=600)
            sys.stdout.write(r.stdout if r.returncode == 0 else "%s: FAILED\n%s\n" % (item, (r.stdout + r.stderr)[-1500:]))
            sys.stdout.flush()
