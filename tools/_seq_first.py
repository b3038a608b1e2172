import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from __graft_entry__ import load_package
mm = load_package()
N, BLOCK = 4 << 30, 524288
eng = mm.Engine(0)
eng.alloc(N)
mm.synth.RomSpec(42, N, "relativesrch", 1, None, False, BLOCK).apply_device(eng)
eng.download(0, 16)
for item in ["relativesrch", "qzvk", "qzv", "qz*k", "q**k**x", "qz**mb", "ab*de", "a*cd*f", "aaaa", "abcd", "qz", "q*v", "textsrch:2", "qz:2", "q" * 2 + "zvkmbxw" * 9]:
    kw, _, elem = item.partition(":")
    elem = int(elem or 1)
    plan = mm.plan_relative(elem, kw, ord("*") if "*" in kw else 0)
    w = []
    for i in range(4):
        t0 = time.perf_counter()
        r = eng.scan(plan, block_bytes=BLOCK, cap=1 << 25)
        w.append((time.perf_counter() - t0) * 1e3)
    print("%-14s %s path %d parts %d matches %d" % (item[:14], " ".join("%8.3f" % x for x in w), eng.counters()["path"], eng.timings().get("parts", 0), len(r)), flush=True)
