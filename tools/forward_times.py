# SPDX-License-Identifier: GPL-3.0-or-later
"""Dev probe: the forward engine (mmh_set_engine(ctx, 2)) on 1 GiB, device time per scan (timings()["total_ms"]) and wall time:
8-bit plain / wildcard keyword, 16-bit LE / BE, a long keyword, with and without the sweep (MMOORE_FORWARD_SWEEP=0 in a
second run).  -> profiles/rNN_forward_times.log"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package

mm = load_package()
eng = mm.Engine(0)
n = 1 << 30
print("# forward engine, 1 GiB, 512 KiB blocks; MMOORE_FORWARD_SWEEP=%s" % os.environ.get("MMOORE_FORWARD_SWEEP", "(unset: on)"))
for elem, kw, wc, be in ((1, "relativesrch", 0, False), (1, "re*ative*ear*hxy", ord("*"), False), (2, "textsrch", 0, False),
                         (2, "textsrch", 0, True), (2, "te*tsr*h", ord("*"), False), (1, "a quite long keyword of forty-one symbols", 0, False)):
    spec = mm.synth.RomSpec(42, n, kw if len(kw) <= 32 else kw[:12], elem, wc or None, be)
    eng.alloc(n)
    spec.apply_device(eng)
    plan = mm.plan_relative(elem, kw, wc)
    eng.set_engine(0)
    want = eng.scan(plan, block_bytes=524288, big_endian=be)
    eng.set_engine(2)
    dev, wall = [], []
    for _ in range(6):
        t0 = time.perf_counter()
        got = eng.scan(plan, block_bytes=524288, big_endian=be)
        wall.append((time.perf_counter() - t0) * 1e3)
        dev.append(eng.timings()["total_ms"])
    eng.set_engine(0)
    assert got.tolist() == want.tolist(), kw
    print("%d-bit %s %-44s matches %6d  path %d | device %.3f ms (best %.3f)  wall %.3f ms" % (
        8 * elem, "BE" if be else "LE", "'" + kw + "'", len(got), eng.counters()["path"], sorted(dev)[len(dev) // 2], min(dev), min(wall)), flush=True)
