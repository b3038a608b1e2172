# SPDX-License-Identifier: GPL-3.0-or-later
"""Dev probe: short keywords (many candidates) -- per-candidate resolvers vs the dense engine."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
mm = load_package()
eng = mm.Engine(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else (1 << 30)
eng.alloc(n)
eng.synth(42)
for elem, kw in ((1, "the"), (1, "hi"), (1, "cake"), (1, "c*ke"), (2, "hi")):
    plan = mm.plan_relative(elem, kw, ord("*"))
    for engine in (0, 2):
        eng.set_engine(engine)
        for block in (524288, 0):
            f, t = [], []
            for i in range(6):
                r = eng.scan(plan, block_bytes=block, cap=1 << 22)
                tm = eng.timings(); f.append(tm["filter_ms"]); t.append(tm["total_ms"])
            print("u%-2d %-5s engine %d block %-7d matches %8d filter %.3f ms total %.3f ms %s %s" % (
                elem * 8, kw, engine, block, len(r), min(f), min(t), eng.counters(), mm.filter_shape(plan)["verify_in_filter"]))
eng.set_engine(0)
