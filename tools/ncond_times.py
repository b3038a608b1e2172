# SPDX-License-Identifier: GPL-3.0-or-later
"""Dev probe: streaming-kernel time of the C2 keyword with 1..4 SWAR conditions (MMOORE_FILTER_MAXCOND)."""
import sys, os, subprocess
if len(sys.argv) > 1:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from __graft_entry__ import load_package
    mm = load_package()
    eng = mm.Engine(0)
    n = 4 << 30
    eng.alloc(n); eng.synth(42)
    for elem, kw in ((1, "relativesrch"), (2, "textsrch")):
        plan = mm.plan_relative(elem, kw)
        for i in range(150):
            eng.scan(plan, block_bytes=524288, cap=1 << 20)
        f, t = eng.timing_history(40)
        print("maxcond %s u%d: filter %.4f ms (%.0f GB/s) total %.4f ms %s %s" % (
            os.environ.get("MMOORE_FILTER_MAXCOND"), elem * 8, sum(f) / len(f), n / (sum(f) / len(f)) / 1e6, sum(t) / len(t),
            eng.counters(), mm.filter_shape(plan)["conditions"]))
else:
    for k in (4, 3, 2, 1):
        env = dict(os.environ, MMOORE_FILTER_MAXCOND=str(k))
        subprocess.run([sys.executable, __file__, "child"], env=env)
