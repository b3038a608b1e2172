# SPDX-License-Identifier: GPL-3.0-or-later
"""Dev probe: what appending candidates costs the streaming kernel (4 GiB random bytes + PLANTS keyword plants per MiB).
Run once per setting of MMOORE_BUCKETS / MMOORE_EXP (read once per process)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package

mm = load_package()
PIECE = 256 << 20
per_mib = int(os.environ.get("PLANTS", "16"))
rng = np.random.default_rng(7)
rom = rng.integers(0, 256, PIECE, dtype=np.uint8)
kw = np.frombuffer(b"relativesrch", np.uint8).astype(np.int64)
n = (PIECE >> 20) * per_mib
if n:
    pos = np.sort(rng.choice((PIECE - 64) // 32, size=n, replace=False)) * 32 + rng.integers(0, 16, n)
    shift = rng.integers(-int(kw.min()), 256 - int(kw.max()), n)
    for j, v in enumerate(kw):
        rom[pos + j] = (v + shift).astype(np.uint8)
eng = mm.Engine(0)
eng.alloc(16 * PIECE)
for k in range(16):
    eng.poke(k * PIECE, rom)
plan = mm.plan_relative(1, "relativesrch")
import time
f, t, wall = [], [], []
for i in range(40):
    t0 = time.perf_counter()
    offs = eng.scan(plan, block_bytes=524288, cap=1 << 20)
    w = time.perf_counter() - t0
    tm = eng.timings()
    if i >= 15:
        f.append(tm["filter_ms"])
        t.append(tm["total_ms"])
        wall.append(w * 1e3)
print("BUCKETS=%s EXP=%s DIRECT=%s plants/MiB %d: filter %.4f ms (min %.4f)  device %.4f ms  caller %.4f ms (min %.4f) = %.3f of peak  %s  results %d" % (
    os.environ.get("MMOORE_BUCKETS", "1"), os.environ.get("MMOORE_EXP", "0"), os.environ.get("MMOORE_DIRECT_PUBLISH", "-"), per_mib, np.mean(f), np.min(f),
    np.mean(t), np.mean(wall), np.min(wall), (4 << 30) / np.mean(wall) / 1e6 / 8000, eng.counters(), len(offs)))
