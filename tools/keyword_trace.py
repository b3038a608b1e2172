# SPDX-License-Identifier: GPL-3.0-or-later
"""Dev probe (round 5): five synchronous scans of ONE keyword on C2's 4 GiB ROM, for the library's traces:
    MMOORE_SPLIT_TRACE=1 MMOORE_LANE_TRACE=1 python tools/keyword_trace.py qzv"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
mm = load_package()
eng = mm.Engine(0)
N, BLOCK = 4 << 30, 524288
eng.alloc(N)
mm.synth.RomSpec(42, N, "relativesrch", 1, None, False, BLOCK).apply_device(eng)
kw = sys.argv[1]
if os.environ.get("MM_PLANTS"):                              # random bytes + that many planted matches of the keyword per MiB instead
    rng = np.random.default_rng(2026)
    piece = rng.integers(0, 256, 256 << 20, dtype=np.uint8)
    k = np.frombuffer(kw.replace("*", "a").encode(), np.uint8).astype(np.int64)
    m = (piece.size >> 20) * int(os.environ["MM_PLANTS"])
    pos = np.sort(rng.choice((piece.size - 64) // 32, size=m, replace=False)) * 32 + rng.integers(0, 16, m)
    sh = rng.integers(-int(k.min()), 256 - int(k.max()), m)
    for j, v in enumerate(k):
        piece[pos + j] = (v + sh).astype(np.uint8)
    for i in range(16):
        eng.poke(i * (256 << 20), piece)
plan = mm.plan_relative(1, kw, ord("*") if "*" in kw else 0)
for i in range(5):
    sys.stderr.write("---- scan %d\n" % i)
    t0 = time.perf_counter(); r = eng.scan(plan, block_bytes=BLOCK, cap=1 << 22)
    sys.stderr.write("---- %.3f ms, %d matches %s %s\n" % ((time.perf_counter() - t0) * 1e3, len(r), eng.timings(), eng.counters()))
