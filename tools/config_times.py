# SPDX-License-Identifier: GPL-3.0-or-later
"""Dev probe: stage timings of the BASELINE configs at full size (no oracle check)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
mm = load_package()
eng = mm.Engine(0)
cfgs = [("C2 u8 L12 4GiB", 4 << 30, "relativesrch", 1, None, False),
        ("C3 u8 L16 3wc 4GiB", 4 << 30, "re*ative*ear*hxy", 1, ord("*"), False),
        ("C4 u16 LE L8 8GiB", 8 << 30, "textsrch", 2, None, False),
        ("C4' u16 BE L8 8GiB", 8 << 30, "textsrch", 2, None, True)]
for name, n, kw, elem, wc, be in cfgs:
    spec = mm.synth.RomSpec(42, n, kw, elem, wc, be)
    eng.alloc(n)
    spec.apply_device(eng)
    plan = mm.plan_relative(elem, kw, wc or 0)
    # the streaming kernel alone: ONE launch over the whole ROM per scan (MMH_ROUTE_NO_SPLIT) ...
    eng.set_route(mm.ROUTE_NO_SPLIT)
    f, t = [], []
    for i in range(60):
        r = eng.scan(plan, block_bytes=524288, big_endian=be)
        tm = eng.timings(); f.append(tm["filter_ms"]); t.append(tm["total_ms"])
    eng.set_route(0)
    # ... and what a caller of mmh_scan gets (a pipeline of parts on ROMs of this size): wall time per scan
    for i in range(10):
        eng.scan(plan, block_bytes=524288, big_endian=be)
    t0 = time.perf_counter()
    for i in range(40):
        r = eng.scan(plan, block_bytes=524288, big_endian=be)
    sync_ms = (time.perf_counter() - t0) / 40 * 1e3
    parts = eng.timings()["parts"]
    # the same scans with three tickets outstanding (mmh_scan_submit / mmh_scan_collect): wall time per scan
    def in_flight(n):
        tickets, last = [], None
        t0 = time.perf_counter()
        for _ in range(n):
            tickets.append(eng.submit(plan, block_bytes=524288, big_endian=be))
            if len(tickets) == 3:
                last = eng.collect(tickets.pop(0))
        while tickets:
            last = eng.collect(tickets.pop(0))
        return (time.perf_counter() - t0) / n * 1e3, last
    in_flight(30)
    per_scan, last = in_flight(200)
    assert len(last) == len(r) and (last == r).all()
    k = 20
    print("%-22s matches %6d  streaming kernel alone %.3f ms (%.0f GB/s), device time of a one-launch scan %.3f ms | mmh_scan for the caller %.3f ms "
          "(%.0f GB/s, %d parts) | in flight %.3f ms per scan (%.0f GB/s)  %s" % (
              name, len(r), sum(f[-k:]) / k, n / (sum(f[-k:]) / k) / 1e6, sum(t[-k:]) / k, sync_ms, n / sync_ms / 1e6, parts, per_scan, n / per_scan / 1e6,
              eng.counters()))
