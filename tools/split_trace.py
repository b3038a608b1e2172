# SPDX-License-Identifier: GPL-3.0-or-later
"""Dev probe (round 5): where the parts of a split scan lie in time (MMOORE_LANE_TRACE=1 prints every part's streaming and tail
kernel against a common origin) -- the text-like ROM, one keyword.   MMOORE_LANE_TRACE=1 python tools/split_trace.py 'th*s'"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
kw = sys.argv[1] if len(sys.argv) > 1 else "th*s"
sys.argv = sys.argv[:1]
from __graft_entry__ import load_package  # noqa: E402

mm = load_package()
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "candidate_density.py")).read().split("eng = mm.Engine(0)")[0]
ns = {"__file__": __file__}
exec(compile(src, "candidate_density.py", "exec"), ns)
rng = np.random.default_rng(2026)
for per_mib in (1, 4, 16, 64, 256, 4096):
    ns["plant"](ns["random_piece"](rng), rng, "relativesrch", per_mib)
rom = ns["text_like_piece"](rng)
eng = mm.Engine(0)
PIECE = 256 << 20
eng.alloc(16 * PIECE)
if kw == "C2":                                                # bench.py's ROM and keyword instead of the text-like ROM
    kw = "relativesrch"
    mm.synth.RomSpec(42, 16 * PIECE, kw, 1, None, False, 524288).apply_device(eng)
else:
    for k in range(16):
        eng.poke(k * PIECE, rom)
plan = mm.plan_relative(1, kw, ord("*") if "*" in kw else 0)
for i in range(6):
    sys.stderr.write("---- scan %d\n" % i)
    t0 = time.perf_counter()
    offs = eng.scan(plan, block_bytes=524288, cap=1 << 20)
    sys.stderr.write("---- scan %d: %.3f ms for the caller, %d matches, %s %s\n" % (i, (time.perf_counter() - t0) * 1e3, len(offs), eng.timings(), eng.counters()))
