# SPDX-License-Identifier: GPL-3.0-or-later
"""Dev probe: the reference benchmark's keyword shapes (bench_search cases) on an HBM-resident
random buffer -- stage timings and which engine path each one takes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
mm = load_package()
eng = mm.Engine(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else (256 << 20)
eng.alloc(n)
eng.synth(42)
cases = [("plain", "abcde", 0), ("front", "*bcde", ord("*")), ("middle", "ab*de", ord("*")), ("back", "abcd*", ord("*")),
         ("mid2", "ab*de*gh", ord("*")), ("long-mid", "abc*efgh*jkl", ord("*"))]
for elem in (1, 2):
    for name, kw, wc in cases:
        plan = mm.plan_relative(elem, kw, wc)
        for block in (0, 524288):
            f, t = [], []
            for i in range(12):
                r = eng.scan(plan, block_bytes=block)
                tm = eng.timings(); f.append(tm["filter_ms"]); t.append(tm["total_ms"])
            k = 6
            print("u%-2d %-9s block %-7d matches %7d filter %.3f ms total %.3f ms (%.0f GB/s) %s" % (
                elem * 8, name, block, len(r), sum(f[-k:]) / k, sum(t[-k:]) / k, n / (sum(t[-k:]) / k) / 1e6, eng.counters()))
