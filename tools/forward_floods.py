# SPDX-License-Identifier: GPL-3.0-or-later
"""Dev probe: the forward engine forced (mmh_set_engine(ctx, 2)) on 1 GiB of data where nearly every tile has something to
report -- 4096 plants per MiB, alphabets of 3 / 16 symbols -- and on the same data with a keyword that never matches:
what the sparse sweep's reading costs where it cannot save anything, what it saves on low-entropy data.  Run it again
with MMOORE_FORWARD_SWEEP=0 for the engine without the sweep.  -> profiles/rNN_forward_floods.log"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
import numpy as np
mm = load_package()
eng = mm.Engine(0)
PIECE = 256 << 20
rng = np.random.default_rng(2026)
eng.alloc(4 * PIECE)
def plant(rom, kw, per_mib):
    k = np.frombuffer(kw.encode(), np.uint8).astype(np.int64)
    n = (rom.size >> 20) * per_mib
    pos = np.sort(rng.choice((rom.size - 64) // 32, size=n, replace=False)) * 32 + rng.integers(0, 16, n)
    sh = rng.integers(-int(k.min()), 256 - int(k.max()), n)
    for j, v in enumerate(k):
        rom[pos + j] = (v + sh).astype(np.uint8)
for label, per_mib, alpha in (("random + 4096 plants / MiB", 4096, 256), ("alphabet of 3 symbols", 0, 3), ("alphabet of 16 symbols", 0, 16)):
    rom = rng.integers(0, alpha, PIECE, dtype=np.uint8)
    if per_mib:
        plant(rom, "relativesrch", per_mib)
    for k in range(4):
        eng.poke(k * PIECE, rom)
    for kw in ("relativesrch", "abc"):
        plan = mm.plan_relative(1, kw)
        eng.set_engine(2)
        dev = []
        for _ in range(4):
            r = eng.scan(plan, block_bytes=524288, cap=1 << 20)
            dev.append(eng.timings()["total_ms"])
        eng.set_engine(0)
        print("%-28s '%s' 1 GiB forced forward engine: matches %9d  device %.3f ms (best %.3f)  sweep=%s" % (label, kw, len(r), sorted(dev)[2], min(dev), os.environ.get("MMOORE_FORWARD_SWEEP", "on")), flush=True)
