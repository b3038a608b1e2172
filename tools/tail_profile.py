# SPDX-License-Identifier: GPL-3.0-or-later
"""Dev probe: the tail kernel at tens of thousands of candidates (a text-like 1 GiB ROM, 'water'), enough synchronous scans
for rocprofv3 averages.  MMOORE_DENSE_SPLIT=0 so that every scan is ONE streaming kernel + ONE tail kernel.
   python3 tools/tail_profile.py [keyword] [scans]"""
import os
import sys
import time

os.environ.setdefault("MMOORE_DENSE_SPLIT", "0")
os.environ.setdefault("MM_DENSITY_PIECES", "4")
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
src = open(os.path.join(HERE, "candidate_density.py")).read()
exec(src[:src.index("def measure(")])                      # the ROM generators (text_like_piece, ...) and `mm`
kw = sys.argv[1] if len(sys.argv) > 1 else "water"
scans = int(sys.argv[2]) if len(sys.argv) > 2 else 40
eng = mm.Engine(0)
n = NPIECES * PIECE
eng.alloc(n)
rom = text_like_piece(np.random.default_rng(2026))
for k in range(NPIECES):
    eng.poke(k * PIECE, rom)
plan = mm.plan_relative(1, kw, ord("*") if "*" in kw else 0)
for _ in range(5):
    offs = eng.scan(plan, block_bytes=524288, cap=1 << 20)
f, t = [], []
for _ in range(scans):
    offs = eng.scan(plan, block_bytes=524288, cap=1 << 20)
    tm = eng.timings()
    f.append(tm["filter_ms"]); t.append(tm["total_ms"])
c = eng.counters()
print("'%s' on a text-like %d MiB ROM: %d candidates, %d matches, %d look-back windows (%.2f per candidate), path %d, streaming kernel %.3f ms, behind it %.3f ms" % (
    kw, n >> 20, c["candidates"], len(offs), c["tiles_walked"], c["tiles_walked"] / max(1, c["candidates"]), c["path"], float(np.mean(f)),
    float(np.mean(t)) - float(np.mean(f))))
