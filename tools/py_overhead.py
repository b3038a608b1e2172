# SPDX-License-Identifier: GPL-3.0-or-later
"""Dev probe: host wall time per scan from Python (ctypes) vs device time, with / without torch in the process."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--torch" in sys.argv:
    import torch
    torch.cuda.set_device(0)
    buf = torch.empty((4 << 30) + 32, dtype=torch.uint8, device="cuda")
from __graft_entry__ import load_package
mm = load_package()
eng = mm.Engine(0)
n = 4 << 30
if "--torch" in sys.argv:
    eng.attach(buf.data_ptr(), n)
else:
    eng.alloc(n)
mm.synth.RomSpec(42, n, "relativesrch", 1).apply_device(eng)
plan = mm.plan_relative(1, "relativesrch")
for i in range(300):
    eng.scan(plan, block_bytes=524288)
t0 = time.perf_counter()
K = 60
for i in range(K):
    r = eng.scan(plan, block_bytes=524288)
wall = (time.perf_counter() - t0) / K
f, t = eng.timing_history(K)
print("torch" if "--torch" in sys.argv else "plain", "wall/scan %.1f us device %.1f us overhead %.1f us" % (wall * 1e6, t.mean() * 1e3, wall * 1e6 - t.mean() * 1e3))
