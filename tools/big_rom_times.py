# SPDX-License-Identifier: GPL-3.0-or-later
"""Dev probe: one scan of a 64 GiB / 128 GiB ROM resident in one GPU's HBM."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
mm = load_package()
eng = mm.Engine(0)
for gib in (64, 128, 200):
    n = gib << 30
    try:
        eng.alloc(n)
    except mm.MMError as e:
        print(gib, "GiB: allocation failed:", e); continue
    mm.synth.RomSpec(42, n, "relativesrch", 1, plants_per_mib=0).apply_device(eng)
    plan = mm.plan_relative(1, "relativesrch")
    for i in range(12):
        r = eng.scan(plan, block_bytes=524288)
    f, t = eng.timing_history(8)
    print("%d GiB: %d matches, filter %.3f ms (%.0f GB/s), scan %.3f ms (%.0f GB/s)" % (gib, len(r), f.mean(), n / f.mean() / 1e6, t.mean(), n / t.mean() / 1e6), flush=True)
