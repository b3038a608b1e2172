# SPDX-License-Identifier: GPL-3.0-or-later
"""Dev probe: what a synchronous mmh_scan costs its caller beyond the device time, from Python (ctypes, bench.py's way);
tools/scan_probe.bin prints the same from C++.  4 GiB bench ROM, 'relativesrch'."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
mm = load_package()
t0 = time.perf_counter()
eng = mm.Engine(0)
t1 = time.perf_counter()
eng2 = mm.Engine(0)
t2 = time.perf_counter()
eng2.close()
print("first mmh_create of the process (HIP initialisation + the known-answer self-test through every route): %.1f ms; the next one: %.2f ms" % (
    (t1 - t0) * 1e3, (t2 - t1) * 1e3))
n = 4 << 30
eng.alloc(n)
mm.synth.RomSpec(42, n, "relativesrch", 1, None, False, 524288).apply_device(eng)
plan = mm.plan_relative(1, "relativesrch")
for _ in range(300):
    eng.scan(plan, block_bytes=524288)
for label, reps in (("python", 400),):
    t0 = time.perf_counter()
    for _ in range(reps):
        offs = eng.scan(plan, block_bytes=524288)
    wall = (time.perf_counter() - t0) / reps * 1e3
    f, t = eng.timing_history(64)
    print("%s: wall %.4f ms per scan, device total %.4f ms (streaming kernel %.4f), caller's overhead %.1f us, %d matches" % (
        label, wall, float(np.mean(t)), float(np.mean(f)), (wall - float(np.mean(t))) * 1e3, len(offs)))
print(eng.health())
