# SPDX-License-Identifier: GPL-3.0-or-later
"""Dev probe: per-stage device times of mmh_scan for ROMs with 0 / few / many plants."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
import numpy as np
mm = load_package()
eng = mm.Engine(0)
plan = mm.plan_relative(1, "relativesrch")
n = 1 << 30
eng.alloc(n)
for label, per_mib, runs in (("no plants", 0, False), ("1/MiB", 1, False), ("1/MiB + runs", 1, True), ("8/MiB", 8, False)):
    spec = mm.synth.RomSpec(42, n, "relativesrch", 1, plants_per_mib=max(per_mib, 1), runs=runs)
    if per_mib == 0:
        eng.synth(42, 0)
    else:
        spec.apply_device(eng)
    for _ in range(3):
        r = eng.scan(plan, block_bytes=524288)
    print(label, len(r), eng.timings(), eng.counters())
