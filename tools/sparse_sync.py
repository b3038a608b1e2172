# SPDX-License-Identifier: GPL-3.0-or-later
"""Dev probe (round 5): wall time of a synchronous mmh_scan of C2 / C3 / C4 scanned again and again (known sparse), for the
knobs of the split pipeline's sparse schedule.   MMOORE_SPARSE_TAILPIECE_MIB=128 python tools/sparse_sync.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

mm = load_package()
eng = mm.Engine(0)
for name, gib, elem, kw, wc, be in (("C2", 4, 1, "relativesrch", 0, False), ("C3", 4, 1, "re*ative*ear*hxy", ord("*"), False), ("C4", 8, 2, "textsrch", 0, False)):
    n = gib << 30
    eng.alloc(n)
    mm.synth.RomSpec(42, n, kw, elem, wc or None, be, 524288).apply_device(eng)
    plan = mm.plan_relative(elem, kw, wc)
    for _ in range(30):
        eng.scan(plan, block_bytes=524288, big_endian=be)
    reps = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(40):
            offs = eng.scan(plan, block_bytes=524288, big_endian=be)
        reps.append((time.perf_counter() - t0) / 40 * 1e3)
    print("%s: synchronous mmh_scan %.4f ms median of 5 x 40 (min %.4f), %d parts, %d matches; env %s" % (
        name, float(np.median(reps)), min(reps), eng.timings()["parts"], len(offs),
        " ".join("%s=%s" % kv for kv in sorted(os.environ.items()) if kv[0].startswith("MMOORE_")) or "(defaults)"), flush=True)
