# SPDX-License-Identifier: GPL-3.0-or-later
"""Dev probe (round 5): ONE forward-engine case, 30 forced scans of 1 GiB -- for rocprofv3 --pmc averages per case
(tools/forward_counters.sh).   python tools/forward_one.py flood4096|alpha3|alpha16|plain8|wild8|plain16|long41"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402

mm = load_package()
case = sys.argv[1] if len(sys.argv) > 1 else "flood4096"
eng = mm.Engine(0)
n = 1 << 30
PIECE = 256 << 20
rng = np.random.default_rng(2026)
elem, kw, wc, be = 1, "relativesrch", 0, False
if case in ("flood4096", "alpha3", "alpha16"):
    alpha = {"flood4096": 256, "alpha3": 3, "alpha16": 16}[case]
    rom = rng.integers(0, alpha, PIECE, dtype=np.uint8)
    if case == "flood4096":
        k = np.frombuffer(kw.encode(), np.uint8).astype(np.int64)
        m = (rom.size >> 20) * 4096
        pos = np.sort(rng.choice((rom.size - 64) // 32, size=m, replace=False)) * 32 + rng.integers(0, 16, m)
        sh = rng.integers(-int(k.min()), 256 - int(k.max()), m)
        for j, v in enumerate(k):
            rom[pos + j] = (v + sh).astype(np.uint8)
    else:
        kw = "abc"
    eng.alloc(n)
    for i in range(4):
        eng.poke(i * PIECE, rom)
else:
    elem, kw, wc, be = {"plain8": (1, "relativesrch", 0, False), "wild8": (1, "re*ative*ear*hxy", ord("*"), False),
                        "plain16": (2, "textsrch", 0, False), "long41": (1, "a quite long keyword of forty-one symbols", 0, False)}[case]
    eng.alloc(n)
    mm.synth.RomSpec(42, n, kw if len(kw) <= 32 else kw[:12], elem, wc or None, be).apply_device(eng)
plan = mm.plan_relative(elem, kw, wc)
eng.set_engine(2)
dev = []
for _ in range(30):
    r = eng.scan(plan, block_bytes=524288, big_endian=be, cap=1 << 23)
    dev.append(eng.timings()["total_ms"])
print("%s: %d-bit '%s' 1 GiB forced forward engine: %d matches, device %.3f ms median (sweep %s)" % (
    case, 8 * elem, kw, len(r), float(np.median(dev)), os.environ.get("MMOORE_FORWARD_SWEEP", "on")))
