# SPDX-License-Identifier: GPL-3.0-or-later
"""Dev probe (round 6): the streaming kernel's OWN duration (HIP events on its dispatch, mmh_last_timings) per keyword on
C2's 4 GiB ROM, one launch over the whole ROM (MMH_ROUTE_NO_SPLIT) -- what a filter shape costs, apart from the tail
kernel and the caller's wait.
    python tools/keyword_kernel_ms.py [keyword[:elem] ...]      -> stdout (profiles/r06_wide_shapes.log)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

mm = load_package()
eng = mm.Engine(0)
N, BLOCK = 4 << 30, 524288
eng.alloc(N)
mm.synth.RomSpec(42, N, "relativesrch", 1, None, False, BLOCK).apply_device(eng)
eng.set_route(16)
DEFAULT = ["relativesrch", "mo*ke", "ab*defgh", "q*v*m*x", "qz*k", "q*vk", "qz**mb", "qzv**mb", "qz***mb*x", "qzk**mb**x", "q**k**xw", "q***k***xw", "qz*k*mbx",
           "textsrch:2", "q*v*m:2", "qz**mb:2", "q***k**x:2"]
for item in (sys.argv[1:] or DEFAULT):
    kw, _, elem = item.partition(":")
    elem = int(elem or 1)
    plan = mm.plan_relative(elem, kw, ord("*") if "*" in kw else 0)
    shape = mm.filter_shape(plan)
    ks, ts = [], []
    for i in range(24):
        r = eng.scan(plan, block_bytes=BLOCK, cap=1 << 22)
        t = eng.timings()
        if i >= 4:
            ks.append(t["filter_ms"])
            ts.append(t["total_ms"])
    path = eng.counters()["path"]
    if path == 3 or not np.median(ks):
        # (the scan went to the forward engine -- `ab*defgh` matches the ROM's ramps wholesale: no streaming-kernel time of its own)
        print("%2d-bit %-14s conditions %s shape %3d | forward engine (path %d): no streaming kernel of its own | scan on the device %.4f ms | %d matches" % (
            8 * elem, kw, shape["conditions"], shape["shape"], path, np.median(ts), len(r)), flush=True)
        continue
    print("%2d-bit %-14s conditions %s shape %3d | kernel median %.4f min %.4f ms = %4.0f GB/s | scan on the device %.4f ms | %d matches, %d candidates" % (
        8 * elem, kw, shape["conditions"], shape["shape"], np.median(ks), min(ks), N / np.median(ks) / 1e6, np.median(ts), len(r),
        eng.counters()["candidates"]), flush=True)
