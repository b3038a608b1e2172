# SPDX-License-Identifier: GPL-3.0-or-later
"""Dev probe (round 5): what a synchronous mmh_scan costs the caller over ROM sizes -- a real ROM file is 1-64 MiB, the bench
ROM 4 GiB.  C2's keyword and ROM recipe at every size, 512 KiB blocks; median wall time of 60 scans, the device time of the
scan (HIP events), which route ran.    python tools/size_sweep.py [wildcard]      -> profiles/r05_size_sweep.log"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
wild = len(sys.argv) > 1
sys.argv = sys.argv[:1]
from __graft_entry__ import load_package  # noqa: E402

mm = load_package()
eng = mm.Engine(0)
BLOCK = 524288
kw, wc = ("re*ative*ear*hxy", ord("*")) if wild else ("relativesrch", 0)
print("# synchronous mmh_scan over ROM sizes: 8-bit '%s', 512 KiB blocks, ROM resident in HBM" % kw)
for mib in (1, 4, 16, 64, 256, 512, 1024, 2048, 4096):
    n = mib << 20
    spec = mm.synth.RomSpec(42, n, kw, 1, wc or None, False, BLOCK)
    eng.alloc(n)
    spec.apply_device(eng)
    plan = mm.plan_relative(1, kw, wc)
    for _ in range(10):
        r = eng.scan(plan, block_bytes=BLOCK)
    wall, dev, filt = [], [], []
    for _ in range(60):
        t0 = time.perf_counter()
        r = eng.scan(plan, block_bytes=BLOCK)
        wall.append((time.perf_counter() - t0) * 1e6)
        tm = eng.timings()
        dev.append(tm["total_ms"] * 1e3)
        filt.append(tm["filter_ms"] * 1e3)
    w, d = float(np.median(wall)), float(np.median(dev))
    print("%5d MiB  %5d matches  caller %8.1f us (min %8.1f) = %7.1f GB/s | device time of the scan %8.1f us, streaming kernel(s) %8.1f us, parts %d | %s" % (
        mib, len(r), w, min(wall), n / w / 1e3, d, float(np.median(filt)), eng.timings()["parts"], eng.counters()), flush=True)
