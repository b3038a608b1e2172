import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from __graft_entry__ import load_package
mm = load_package()
BLOCK = 524288
for N in (256 << 20, 1 << 30, 4 << 30):
    eng = mm.Engine(0)
    eng.alloc(N)
    mm.synth.RomSpec(42, N, "relativesrch", 1, None, False, BLOCK).apply_device(eng)
    eng.set_route(16)
    for kw in ("abcd", "aaaa", "relativesrch"):
        plan = mm.plan_relative(1, kw, 0)
        for engine in (0, 2):
            eng.set_engine(engine)
            w = []
            for i in range(5):
                t0 = time.perf_counter()
                r = eng.scan(plan, block_bytes=BLOCK, cap=1 << 22)
                w.append((time.perf_counter() - t0) * 1e3)
            print("%5d MiB %-12s engine %d: wall %s | timings %s | path %d matches %d" % (N >> 20, kw, engine, " ".join("%.3f" % x for x in w), eng.timings(), eng.counters()["path"], len(r)), flush=True)
    eng.close()
