# SPDX-License-Identifier: GPL-3.0-or-later
"""Dev probe (round 5): scans in flight (three tickets) with and without the events at the scans' starts (mmh_set_timing): no
difference -- 0.6820-0.6827 ms per 4 GiB scan either way; the event only costs where a dispatch is on the caller's critical path."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
mm = load_package()
eng = mm.Engine(0)
n = 4 << 30
eng.alloc(n)
mm.synth.RomSpec(42, n, "relativesrch", 1, None, False, 524288).apply_device(eng)
plan = mm.plan_relative(1, "relativesrch", 0)
def in_flight(k):
    tickets = []
    t0 = time.perf_counter()
    for _ in range(k):
        tickets.append(eng.submit(plan, block_bytes=524288))
        if len(tickets) == 3:
            eng.collect(tickets.pop(0))
    while tickets:
        eng.collect(tickets.pop(0))
    return (time.perf_counter() - t0) / k * 1e3
in_flight(50)
for rep in range(3):
    for on in (True, False):
        eng.set_timing(on)
        in_flight(20)
        print("timing events %s: %.4f ms per scan in flight (400 scans)" % ("on " if on else "off", in_flight(400)), flush=True)
