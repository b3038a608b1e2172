# SPDX-License-Identifier: GPL-3.0-or-later
"""Dev probe (round 5): ONE keyword on the text-like ROM of tools/candidate_density.py, 30 one-launch scans (MMH_ROUTE_NO_SPLIT)
-- for rocprofv3 --pmc averages of the streaming kernel with its rare path busy (tools/dense_counters.sh).
    python tools/dense_one.py 'th*s'"""
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
kw = sys.argv[1] if len(sys.argv) > 1 else "th*s"
sys.argv = sys.argv[:1]
os.environ.setdefault("MM_DENSITY_PIECES", "16")
from __graft_entry__ import load_package  # noqa: E402

mm = load_package()
PIECE, BLOCK = 256 << 20, 524288
npieces = int(os.environ["MM_DENSITY_PIECES"])
eng = mm.Engine(0)
eng.alloc(npieces * PIECE)
spec_cd = importlib.util.spec_from_file_location("cd", os.path.join(HERE, "candidate_density.py"))
src = open(spec_cd.origin).read().split("eng = mm.Engine(0)")[0]      # its ROM builders only
ns = {"__file__": spec_cd.origin}
exec(compile(src, spec_cd.origin, "exec"), ns)
rng = np.random.default_rng(2026)
for per_mib in (1, 4, 16, 64, 256, 4096):
    ns["plant"](ns["random_piece"](rng), rng, "relativesrch", per_mib)
rom = ns["text_like_piece"](rng)
for k in range(npieces):
    eng.poke(k * PIECE, rom)
plan = mm.plan_relative(1, kw, ord("*") if "*" in kw else 0)
eng.set_route(mm.ROUTE_NO_SPLIT)
f = []
for _ in range(30):
    r = eng.scan(plan, block_bytes=BLOCK, cap=1 << 20)
    f.append(eng.timings()["filter_ms"])
print("'%s' on the text-like ROM, %d GiB, one launch per scan: %d matches, %s, streaming kernel median %.4f ms" % (
    kw, (npieces * PIECE) >> 30, len(r), eng.counters(), float(np.median(f))))
