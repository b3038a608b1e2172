# SPDX-License-Identifier: GPL-3.0-or-later
"""Dev probe (run under tools/trace_kernels.sh): ~4 K candidates spread over 4 GiB vs packed into 64 MiB."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
mm = load_package()
eng = mm.Engine(0)
n = 4 << 30
eng.alloc(n)
plan = mm.plan_relative(1, "relativesrch")
mm.synth.RomSpec(42, n, "relativesrch", 1, runs=False).apply_device(eng)
for i in range(60):
    r = eng.scan(plan, block_bytes=524288)
print("spread", len(r), eng.counters())
mm.synth.RomSpec(42, 64 << 20, "relativesrch", 1, plants_per_mib=64, runs=False).apply_device(eng)
for i in range(60):
    r = eng.scan(plan, block_bytes=524288)
print("packed", len(r), eng.counters())
