# SPDX-License-Identifier: GPL-3.0-or-later
"""Dev probe: many repetitions of the same scans; every result must equal the first (looks for rare races)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from __graft_entry__ import load_package
mm = load_package()
eng = mm.Engine(0)
n = 1 << 30
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
for elem, kw, wc, be in ((1, "relativesrch", None, False), (1, "re*ative*ear*hxy", ord("*"), False), (2, "textsrch", None, True), (1, "abcde", None, False)):
    eng.alloc(n)
    mm.synth.RomSpec(42, n, kw, elem, wc, be, plants_per_mib=4).apply_device(eng)
    plan = mm.plan_relative(elem, kw, wc or 0)
    first = eng.scan(plan, block_bytes=524288, big_endian=be)
    bad = 0
    t0 = time.perf_counter()
    for i in range(reps):
        r = eng.scan(plan, block_bytes=524288, big_endian=be)
        bad += not np.array_equal(r, first)
    prev = None
    for i in range(reps):
        t = eng.submit(plan, block_bytes=524288, big_endian=be)
        if prev is not None:
            bad += not np.array_equal(eng.collect(prev), first)
        prev = t
    bad += not np.array_equal(eng.collect(prev), first)
    print("%-18s u%d: %d matches, %d + %d scans in %.1f s, %d deviating results, path %d" % (
        kw, 8 * elem, len(first), reps, reps, time.perf_counter() - t0, bad, eng.counters()["path"]), flush=True)
