# SPDX-License-Identifier: GPL-3.0-or-later
"""Dev probe: wall time per scan, synchronous mmh_scan vs two scans in flight (submit / collect)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
mm = load_package()
eng = mm.Engine(0)
n = 4 << 30
eng.alloc(n)
mm.synth.RomSpec(42, n, "relativesrch", 1).apply_device(eng)
plan = mm.plan_relative(1, "relativesrch")
for i in range(300):
    eng.scan(plan, block_bytes=524288)
K = 200
t0 = time.perf_counter()
for i in range(K):
    r = eng.scan(plan, block_bytes=524288)
sync = (time.perf_counter() - t0) / K
for rep in range(2):
    t0 = time.perf_counter()
    prev = None
    for i in range(K):
        t = eng.submit(plan, block_bytes=524288)
        if prev is not None:
            r2 = eng.collect(prev)
        prev = t
    r2 = eng.collect(prev)
    pipe = (time.perf_counter() - t0) / K
    f, tt = eng.timing_history(60)
    print("sync %.1f us/scan (%.0f GB/s) | two in flight %.1f us/scan (%.0f GB/s)  filter %.1f us, scan start-to-end %.1f us  same results %s" % (
        sync * 1e6, n / sync / 1e9, pipe * 1e6, n / pipe / 1e9, f.mean() * 1e3, tt.mean() * 1e3, r.tolist() == r2.tolist()))
