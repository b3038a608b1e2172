# SPDX-License-Identifier: GPL-3.0-or-later
"""Dev probe: BASELINE C4 (8 GiB, 16-bit LE, 8-symbol keyword) alone, for rocprofv3 kernel stats of mm_filter_u16."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
mm = load_package()
eng = mm.Engine(0)
n = 8 << 30
spec = mm.synth.RomSpec(42, n, "textsrch", 2, None, False)
eng.alloc(n); spec.apply_device(eng)
plan = mm.plan_relative(2, "textsrch")
for i in range(60):
    r = eng.scan(plan, block_bytes=524288)
f, t = eng.timing_history(40)
print("C4 u16 LE L8 8 GiB: %d matches, filter %.4f ms, device total %.4f ms" % (len(r), f.mean(), t.mean()))
