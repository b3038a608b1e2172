# SPDX-License-Identifier: GPL-3.0-or-later
"""Dev probe: streaming-kernel time on pure random bytes vs the bench ROM (plants / runs), same process."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
mm = load_package()
eng = mm.Engine(0)
n = 4 << 30
elem, kw = 1, "relativesrch"
plan = mm.plan_relative(elem, kw)
eng.alloc(n); eng.synth(42)
def run(tag):
    for i in range(150):
        r = eng.scan(plan, block_bytes=524288)
    f, t = eng.timing_history(50)
    print("u%d %-22s filter %.4f ms (%.0f GB/s) total %.4f ms matches %d %s" % (elem * 8, tag, sum(f) / len(f), n / (sum(f) / len(f)) / 1e6, sum(t) / len(t), len(r), eng.counters()), flush=True)
run("random")
mm.synth.RomSpec(42, n, kw, elem, None, False, plants_per_mib=0).apply_device(eng)
run("runs + straddlers")
mm.synth.RomSpec(42, n, kw, elem, None, False, runs=False).apply_device(eng)
run("plants only")
mm.synth.RomSpec(42, n, kw, elem, None, False).apply_device(eng)
run("bench ROM")
eng.synth(42)
eng.fill(3 << 20, 1 << 20, 0, 0)
run("zero run only")
eng.synth(42)
eng.fill(7 << 20, 1 << 20, 0, 1)
run("ramp run only")
