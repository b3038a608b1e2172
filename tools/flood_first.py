# SPDX-License-Identifier: GPL-3.0-or-later
"""Dev probe (round 5): keywords that match PADDING wholesale (`aaaa` on runs of 0x00 / 0xFF, `abcd` on a ramp: 0.35-0.7 M matches
inside 3 MiB of C2's 4 GiB ROM) -- first scans (a ROM byte rewritten before each) and repeated ones, with / without the pipeline of
parts and the flood hint.   [MMOORE_DENSE_SPLIT=0] [MMOORE_FLOOD_HINT=0] python tools/flood_first.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
mm = load_package()
eng = mm.Engine(0)
N, BLOCK = 4 << 30, 524288
eng.alloc(N)
mm.synth.RomSpec(42, N, "relativesrch", 1, None, False, BLOCK).apply_device(eng)
byte0 = eng.download(0, 1)
for kw in ("aaaa", "abcd"):
    plan = mm.plan_relative(1, kw, 0)
    first, again = [], []
    for i in range(6):
        eng.poke(0, byte0)
        t0 = time.perf_counter(); r = eng.scan(plan, block_bytes=BLOCK, cap=1 << 22); first.append((time.perf_counter() - t0) * 1e3)
    p1 = eng.counters()["path"]
    for i in range(6):
        t0 = time.perf_counter(); r = eng.scan(plan, block_bytes=BLOCK, cap=1 << 22); again.append((time.perf_counter() - t0) * 1e3)
    print(kw, os.environ.get("MMOORE_DENSE_SPLIT", "split on"), "hint", os.environ.get("MMOORE_FLOOD_HINT", "on"), "first scans", [round(x, 2) for x in first], "path", p1, "| again", [round(x, 2) for x in again], "path", eng.counters()["path"], len(r), "matches")
