# SPDX-License-Identifier: GPL-3.0-or-later
"""Dev probe: the forward (dense) engine alone, enough launches for rocprofv3 averages.
   python3 tools/dense_profile.py [plain|wild] [scans]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
mm = load_package()
which = sys.argv[1] if len(sys.argv) > 1 else "plain"
scans = int(sys.argv[2]) if len(sys.argv) > 2 else 24
eng = mm.Engine(0)
n = 1 << 30
spec = mm.synth.RomSpec(42, n, "relativesrch", 1)
eng.alloc(n); spec.apply_device(eng)
plan = mm.plan_relative(1, "relativesrch") if which == "plain" else mm.plan_relative(1, "re*ative*ear*hxy", ord("*"))
eng.set_engine(2)
best = 1e9
for _ in range(scans):
    t0 = time.perf_counter(); r = eng.scan(plan, block_bytes=524288); best = min(best, time.perf_counter() - t0)
print("%s: 1 GiB dense engine, %d matches, best %.3f ms, device %.3f ms" % (which, len(r), best * 1e3, eng.timings()["total_ms"]))
