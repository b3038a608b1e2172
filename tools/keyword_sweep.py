# SPDX-License-Identifier: GPL-3.0-or-later
"""Dev probe (round 5): what a synchronous mmh_scan costs over KEYWORD CLASSES on one 4 GiB ROM (C2's recipe: random bytes,
a planted match per MiB, 1 MiB runs of 0x00 / 0xFF / a ramp) -- lengths 2 .. 128, wildcards in every place, 8- and 16-bit:
where the cliffs are (few conditions -> candidate floods -> flood paths / forward engine).
    python tools/keyword_sweep.py        -> profiles/r06_keyword_sweep.log
    [MMOORE_TRACE=split] python tools/keyword_sweep.py aaaa 'ab*de' 2:qz     only these (N: = N-byte elements; trace: what the scan's stages did)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

mm = load_package()
eng = mm.Engine(0)
N, BLOCK = 4 << 30, 524288
mm.synth.RomSpec(42, N, "relativesrch", 1, None, False, BLOCK)
eng.alloc(N)
mm.synth.RomSpec(42, N, "relativesrch", 1, None, False, BLOCK).apply_device(eng)
CASES = [(1, "relativesrch"), (1, "qz"), (1, "qzv"), (1, "qzvk"), (1, "qzvkm"), (1, "qzvkmbxw"), (1, "q*v"), (1, "qz*k"), (1, "q*vk"), (1, "mo*ke"),
         (1, "*zvkm"), (1, "qzvk*"), (1, "q*v*m*x"), (1, "q**k**x"), (1, "qz**mb"), (1, "Bu**er"), (1, "qzv**mb"), (1, "q***k***x"), (1, "qz***mb*x"), (1, "ab*de"), (1, "a*cd*f"),
         (1, "Qzvkm"), (1, "qzvkmbxwqzvkmbxwqzvkmbxwqzvkmbxwq"),
         (1, "q" * 2 + "zvkmbxw" * 9), (1, "zvkmbxw" * 18), (1, "aaaa"), (1, "abcd"),
         (2, "qz"), (2, "qzv"), (2, "qzvk"), (2, "textsrch"), (2, "q*vk"), (2, "qz*k"), (2, "q*v*m"), (2, "q**k"), (2, "qz**mb"), (2, "q***k**x")]
if len(sys.argv) > 1:
    CASES = [(int(a[0]), a[2:]) if a[1:2] == ":" and a[0] in "12" else (1, a) for a in sys.argv[1:]]
print("# synchronous mmh_scan over keyword classes: 4 GiB (C2's ROM), 512 KiB blocks; wall ms = median of 5 after 2 warm-up scans")
for elem, kw in CASES:
    wc = ord("*") if "*" in kw else 0
    try:
        plan = mm.plan_relative(elem, kw, wc)
    except Exception as e:                                    # noqa: BLE001
        print("%2d-bit %-36s refused: %s" % (8 * elem, kw[:36], e))
        continue
    shape = mm.filter_shape(plan)
    wall = []
    for i in range(7):
        t0 = time.perf_counter()
        r = eng.scan(plan, block_bytes=BLOCK, cap=1 << 22)
        wall.append((time.perf_counter() - t0) * 1e3)
    w = float(np.median(wall[2:]))
    tm, ctr = eng.timings(), eng.counters()
    print("%2d-bit %-36s L %3d  conditions %d (shape %3d%s)  %9d matches  %9d candidates  path %d  parts %d | caller %8.3f ms = %6.0f GB/s (first scan %8.3f ms)" % (
        8 * elem, kw[:36], len(kw), shape["ncond"], shape["shape"], ", verified in the filter" if shape.get("verify_in_filter") else "",
        len(r), ctr["candidates"], ctr["path"], tm["parts"], w, N / w / 1e6, wall[0]), flush=True)
