# SPDX-License-Identifier: GPL-3.0-or-later
"""Dev probe: the dense (forward) and sequential engines vs the fast path."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
import numpy as np
mm = load_package()
eng = mm.Engine(0)
def t(label, plan, **kw):
    for e, name in ((0, "auto"), (2, "dense")) + (((1, "seq"),) if "--seq" in sys.argv else ()):   # seq on 1 GiB whole = 97 s
        eng.set_engine(e)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); r = eng.scan(plan, **kw); best = min(best, time.perf_counter() - t0)
        print("%-34s %-6s matches %9d  %9.3f ms  path %d" % (label, name, len(r), best * 1e3, eng.counters()["path"]))
    eng.set_engine(0)
n = 1 << 30
spec = mm.synth.RomSpec(42, n, "relativesrch", 1)
eng.alloc(n); spec.apply_device(eng)
t("1 GiB C2-like, blocks 512K", mm.plan_relative(1, "relativesrch"), block_bytes=524288)
t("1 GiB C2-like, whole buffer", mm.plan_relative(1, "relativesrch"))
t("1 GiB wildcards, blocks", mm.plan_relative(1, "re*ative*ear*hxy", ord("*")), block_bytes=524288)
eng.fill(0, n, 7)
t("1 GiB constant, 'aaa', blocks", mm.plan_relative(1, "aaa"), block_bytes=524288)
n2 = 64 << 20
eng.alloc(n2); eng.fill(0, n2, 7)
t("64 MiB constant, 'aaa', whole", mm.plan_relative(1, "aaa"))
